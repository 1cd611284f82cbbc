"""MI355X-native Stokes-surrogate training path (drop-in for the hot path of
agsiddhant/PBML_Mantle_Convection): symmetric-conv U-Net / ConvAE forward + backward and the
finite-difference loss, as hand-written HIP kernels behind a C ABI (libmantle_hip.so), with
the reference's Python module / trainer surface on top.

Modules mirror the reference file names: symmetric_layers_torch, pytorch_networks_convae,
multigpu, train, datasetio, scaler.
"""
__version__ = "0.1.0"
