"""Golden-vector generator for the BENCHMARKED configuration itself: PlaneRecNet_101 train step at B = 8, 480x640 (BASELINE config 3) -- runs ONLY
in the build container (needs /root/reference).

    python tests/golden/make_golden_r101_b8.py

Same weights as make_golden_r101.py (oracle/synth.py, seed 3), the batch of tests/test_r101_train_gpu.py::test_r101_b8_* (seed 21, numpy seed 5):

1. the real reference (shim-imported, CPU fp32): forward, five loss terms, backward;
2. the oracle in fp32: asserted equal to the reference (as in make_golden_r101.py);
3. the oracle in fp64: the yardstick.  A full fp64 gradient set is 230 MB, so the fixture holds, per parameter,
   - the fp64 gradient's norm and the reference's / the fp32 oracle's rel-L2 distance to it on the FULL tensor (the per-parameter spread the
     GPU test scales its bound with),
   - the fp64 gradient at up to NS seeded sample positions (all of it for tensors of <= NS elements), as float32 (2^-24 relative: three
     orders below the tightest bound), and the reference's fp32 gradient at the same positions.
   The GPU test evaluates  |g_hip - g64| / |g64|  over those positions: an unbiased estimate of the full-tensor rel-L2 whose sampling error
   (~1 / sqrt(NS) of itself) is far below the bound's factor.

Writes tests/golden/e2e_r101_b8_480x640.npz.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import ref_shim, synth, model_ref, loss_ref  # noqa: E402

CN = "PlaneRecNet_101_config"
B, H, W = 8, 480, 640
SEED_W, SEED_X, SEED_NP = 3, 21, 5
NS = 2048
SAMPLE_SEED = 777


def sample_index(numel):
    if numel <= NS:
        return torch.arange(numel)
    return torch.randint(0, numel, (NS,), generator=torch.Generator().manual_seed(SAMPLE_SEED + numel % 9973))


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def oracle_step(sd, x, inst, gtd, dtype):
    arch = model_ref.ARCH[CN]
    sdg = {k: (v.to(dtype).clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else
               (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    names = [k for k, v in sdg.items() if v.requires_grad]
    np.random.seed(SEED_NP)
    out = model_ref.forward(sdg, x.to(dtype), arch, training=True)
    ls = loss_ref.joint_loss(*out, inst, gtd)
    grads = torch.autograd.grad(sum(ls.values()).sum(), [sdg[n] for n in names], allow_unused=True)
    return out, ls, dict(zip(names, grads))


def main():
    torch.set_num_threads(8)
    ref = ref_shim.load_reference(CN)
    cfg = ref["config"].cfg
    net = ref["planerecnet"].PlaneRecNet(cfg)
    sd = synth.make_state_dict(CN, seed=SEED_W)
    net.load_state_dict(sd)
    net.train()
    crit = ref["losses"].PlaneRecNetLoss()
    x, inst, gtd = synth.make_batch(B, H, W, seed=SEED_X)

    t0 = time.time()
    np.random.seed(SEED_NP)
    out = net(x)
    rl = crit(net, *out, inst, gtd)
    net.zero_grad()
    sum(rl.values()).sum().backward()
    rgrads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    rl = {k: float(v) for k, v in rl.items()}
    print("reference step: %.1f s" % (time.time() - t0), rl, flush=True)
    del out, net, crit

    t0 = time.time()
    _, l32, g32 = oracle_step(sd, x, inst, gtd, torch.float32)
    print("oracle fp32 step: %.1f s" % (time.time() - t0), {k: float(v) for k, v in l32.items()}, flush=True)
    for k in rl:
        assert abs(rl[k] - float(l32[k])) <= 1e-4 * max(1.0, abs(rl[k])), k
    zero, worst = [], []
    for n, g in rgrads.items():
        if g.norm().item() < 1e-5 and n.endswith(".bias"):
            zero.append(n)
            assert g32[n].norm().item() < 1e-5, n
        else:
            worst.append((rel_l2(g32[n], g), n))
    worst.sort(reverse=True)
    print("oracle fp32 vs reference gradients: largest rel-L2", worst[:5], flush=True)
    assert worst[0][0] < 2e-3, worst[0]
    g32 = {n: g.detach() for n, g in g32.items() if g is not None}

    t0 = time.time()
    _, l64, g64 = oracle_step(sd, x, inst, gtd, torch.float64)
    print("oracle fp64 step: %.1f s" % (time.time() - t0), {k: float(v) for k, v in l64.items()}, flush=True)

    names = sorted(n for n in rgrads if n not in zero)
    fix = {"grad_names": np.array(names), "grad_structurally_zero": np.array(sorted(zero)), "sample_seed": np.array(SAMPLE_SEED), "ns": np.array(NS)}
    for k in rl:
        fix[k] = np.asarray(rl[k])
        fix["fp64_" + k] = np.asarray(float(l64[k]))
    fix["grad_fp64_norm"] = np.array([g64[n].norm().item() for n in names])
    fix["grad_spread_ref_vs_fp64"] = np.array([rel_l2(rgrads[n], g64[n]) for n in names])
    fix["grad_spread_oracle32_vs_fp64"] = np.array([rel_l2(g32[n], g64[n]) for n in names])
    s64, sref, off, o = [], [], [0], 0
    for n in names:
        idx = sample_index(g64[n].numel())
        s64.append(g64[n].detach().flatten()[idx].to(torch.float32).numpy())
        sref.append(rgrads[n].flatten()[idx].numpy())
        o += idx.numel()
        off.append(o)
    fix["sample_offsets"] = np.array(off, dtype=np.int64)
    fix["grad_fp64_samples"] = np.concatenate(s64)
    fix["grad_ref_samples"] = np.concatenate(sref)
    # how well the sampled estimate tracks the full-tensor rel-L2 (reference vs fp64, where both are known here)
    est = np.array([np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30) for a, b in zip(sref, [s.astype(np.float64) for s in s64])])
    ratio = est / np.maximum(fix["grad_spread_ref_vs_fp64"], 1e-12)
    print("sampled / full rel-L2 (reference vs fp64): percentiles 1/50/99 = %s" % np.round(np.percentile(ratio, [1, 50, 99]), 3), flush=True)
    sp = np.maximum(fix["grad_spread_ref_vs_fp64"], fix["grad_spread_oracle32_vs_fp64"])
    print("fp32-vs-fp64 gradient spread on this batch: median %.1e, max %.1e" % (np.median(sp), sp.max()))
    path = os.path.join(HERE, "e2e_r101_b8_480x640.npz")
    np.savez_compressed(path, **fix)
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
