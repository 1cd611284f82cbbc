"""Velocity scaling law (reference scaler.py:4-71): u, v are divided / multiplied by
5 exp(0.180167667 RaQ + 0.4330392 ln FKT - 0.46052953 ln FKP); p, V, T pass through.
Like the reference, u/v arrays are modified IN PLACE and returned."""
import numpy as np


def velocity_scaler(raq, fkt, fkp):
    return np.exp((raq / 10) * 1.80167667 + np.log(fkt) * 0.4330392 + np.log(fkp) * -0.46052953) * 5


def scale_var(x, raq, fkt, fkp, var):
    if var in ("uprev", "vprev"):
        x /= velocity_scaler(raq, fkt, fkp)
    return x


def unscale_var(x, raq, fkt, fkp, var):
    if var in ("uprev", "vprev"):
        x *= velocity_scaler(raq, fkt, fkp)
    return x
