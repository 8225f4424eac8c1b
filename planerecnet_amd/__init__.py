"""planerecnet_amd -- MI355X-native forward/backward hot path of PlaneRecNet.

Host side (Python, mirrors the reference's module interface): config, backbone, dcn, fpn, planerecnet, losses.
Device side: planerecnet_amd/csrc/*.hip -> libprn_hip.so (C ABI in include/prn.h), reached through ops.py.
Importing the package does not load the HIP library; importing `planerecnet_amd.ops` (or anything that
computes) does, and fails loudly if it is missing.
"""
import os as _os

# HIP multiplexes every stream of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training step uses the
# compute stream, the weight-gradient side stream and -- with more than one rank -- RCCL's streams; with 4 queues the
# exchange shared a queue with compute work and a one-rank probe of the whole exchange path cost 6 ms/step (60.5 -> 66.5 ms),
# with 3 (or 5, 6) it costs 1 ms (sweep in DESIGN.md 4.1d).  The runtime reads the variable when it initialises, i.e. at the
# first HIP call, so setting it at package import is early enough; an explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

__version__ = "0.1.0"
