"""Single-process entry point with the flags of the reference's train script
(reference .ipynb_checkpoints/train-checkpoint.py:32-55), delegating to the multigpu trainer:

    python -m pbml_mantle_convection_amd.train -net unet -l 5 -f 16 -r 3 -k 5 -s 1 -p reflect -lt mass \
        -pp 1 -b 4 -ab 10 --synthetic 64 128 506 --epochs 2
"""
from .multigpu import cli

if __name__ == "__main__":
    cli()
