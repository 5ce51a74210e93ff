"""Constants of /root/reference/global_configs.py:1-19 -- as DEFAULTS only.

The reference hard-wires DEVICE = cuda:0 and edits VISUAL_DIM by hand for MOSEI; here the dims are
constructor arguments everywhere (47 MOSI / 35 MOSEI) and the device is taken from the tensors."""
ACOUSTIC_DIM = 74
VISUAL_DIM = 47          # MOSI; MOSEI = 35
TEXT_DIM = 768
XLNET_INJECTION_INDEX = 1
DATASET_DIMS = {"mosi": dict(visual_dim=47, acoustic_dim=74), "mosei": dict(visual_dim=35, acoustic_dim=74)}
