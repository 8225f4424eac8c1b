import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from planerecnet_amd import ops, timer, targets as T
from planerecnet_amd.config import cfg, set_cfg
from planerecnet_amd.losses import PlaneRecNetLoss
from planerecnet_amd.planerecnet import PlaneRecNet
timer.disable_all()
torch.set_num_threads(4)
dev = torch.device("cuda:0")
set_cfg("PlaneRecNet_101_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg); net.init_head_weights(); net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = T.DeviceTargetBuilder(crit)
pf.submit(inst, (480, 640)); pf.submit(inst, (480, 640))
ops.set_wgrad_async(not os.environ.get("NO_ASYNC"))
FIXED = [None]
W = {}
def wrap_sync(cls):
    orig = cls.synchronize
    def sync(self):
        t0, c0 = time.perf_counter(), time.thread_time()
        r = orig(self)
        import traceback
        fr = traceback.extract_stack(limit=3)[0]
        k = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
        e = W.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += time.perf_counter() - t0; e[2] += time.thread_time() - c0
        return r
    cls.synchronize = sync
wrap_sync(torch.cuda.Event)
PH = {}
def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    if os.environ.get("FIXED_TARGETS"):
        if FIXED[0] is None:
            FIXED[0] = pf.get(depths, dev)
        t = FIXED[0]
        t1 = time.perf_counter()
    else:
        t = pf.get(depths, dev, overlap=True)
        t1 = time.perf_counter()
        pf.submit(inst, (480, 640))
    t2 = time.perf_counter()
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    tot = sum(losses.values()).sum()
    t3 = time.perf_counter()
    tot.backward(); ops.wgrad_join()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    for k, v in (("get", t1 - t0), ("submit", t2 - t1), ("fwd+loss", t3 - t2), ("bwd", t4 - t3), ("adam", t5 - t4)):
        PH[k] = PH.get(k, 0.0) + v
for _ in range(10): step()
torch.cuda.synchronize(); W.clear(); PH.clear()
def thread_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[tid] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return out
N = 30
th0 = thread_cpu()
t0, c0 = time.perf_counter(), time.process_time()
for _ in range(N): step()
torch.cuda.synchronize()
print("free-running: %.2f ms/step wall, %.2f ms/step process CPU" % ((time.perf_counter() - t0) / N * 1e3, (time.process_time() - c0) / N * 1e3))
print("host phases (ms/step):", {k: round(v / N * 1e3, 2) for k, v in PH.items()})
for k, (n, w, c) in sorted(W.items(), key=lambda x: -x[1][1]):
    print("event.synchronize at %-24s %5.1f calls/step  %7.2f ms/step wall  %7.2f ms/step thread CPU" % (k, n / N, w / N * 1e3, c / N * 1e3))
th1 = thread_cpu()
rows = sorted(((th1[k][1] - th0.get(k, (None, 0.0))[1], th1[k][0], k) for k in th1), reverse=True)
print("CPU per OS thread over the %d steps (ms/step):" % N)
for d, comm, tid in rows[:10]:
    print("   %-24s tid %-8s %7.2f%s" % (comm, tid, d / N * 1e3, "   <- main" if int(tid) == os.getpid() else ""))
import threading
print("python threads:", [t.name for t in threading.enumerate()])
print("os threads:", len(os.listdir("/proc/self/task")))
pf.close()
