"""planerecnet_amd -- MI355X-native forward/backward hot path of PlaneRecNet.

Host side (Python, mirrors the reference's module interface): config, backbone, dcn, fpn, planerecnet, losses.
Device side: planerecnet_amd/csrc/*.hip -> libprn_hip.so (C ABI in include/prn.h), reached through ops.py.
Importing the package does not load the HIP library; importing `planerecnet_amd.ops` (or anything that
computes) does, and fails loudly if it is missing.
"""
__version__ = "0.1.0"
