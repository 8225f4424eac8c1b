"""Which OS threads does a training process have, when are they created, and which of them burn CPU?  (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def tids(): return set(os.listdir("/proc/self/task"))
def cpu(tid):
    try:
        f = open("/proc/self/task/%s/stat" % tid).read(); rest = f[f.rindex(")") + 2:].split()
        return (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK")
    except Exception: return 0.0
seen = tids()
def stage(name):
    global seen
    now = tids(); new = sorted(now - seen, key=int); seen = now
    print("%-46s +%d threads %s" % (name, len(new), " ".join(new[:12])), flush=True)
    return new
import torch
stage("import torch")
dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev); torch.cuda.synchronize()
stage("first device op")
s2 = torch.cuda.Stream(); 
with torch.cuda.stream(s2): y = x + 1
torch.cuda.synchronize()
stage("second stream")
ev = torch.cuda.Event(); ev.record(); s2.wait_event(ev); torch.cuda.synchronize()
stage("event record + wait_event")
evb = torch.cuda.Event(blocking=True); evb.record(); evb.synchronize()
stage("blocking event synchronize")
h = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
stage("pin_memory")
d = torch.empty(1 << 20, dtype=torch.uint8, device=dev); d.copy_(h, non_blocking=True); torch.cuda.synchronize()
stage("async H2D from pinned")
h.copy_(d, non_blocking=True); torch.cuda.synchronize()
stage("async D2H to pinned")
w = torch.randn(64, 64, device=dev, requires_grad=True); (w @ w).sum().backward(); torch.cuda.synchronize()
stage("first backward (autograd thread)")
import bench
from planerecnet_amd import ops, timer, targets as T
from planerecnet_amd.config import cfg, set_cfg
from planerecnet_amd.losses import PlaneRecNetLoss
from planerecnet_amd.planerecnet import PlaneRecNet
timer.disable_all(); torch.set_num_threads(4)
set_cfg("PlaneRecNet_101_config"); torch.manual_seed(0)
net = PlaneRecNet(cfg); net.init_head_weights(); net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
stage("model + batch built")
pf = T.DeviceTargetBuilder(crit); pf.submit(inst, (480, 640)); pf.submit(inst, (480, 640)); torch.cuda.synchronize()
stage("DeviceTargetBuilder.submit x2")
ops.set_wgrad_async(True)
def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev, overlap=True); pf.submit(inst, (480, 640))
    out = net(images); losses = crit(net, *out, inst, depths, targets=t)
    sum(losses.values()).sum().backward(); ops.wgrad_join(); opt.step()
step(); torch.cuda.synchronize()
stage("first training step")
for _ in range(8): step()
torch.cuda.synchronize()
stage("eight more steps")
def report(title, fn):
    c0 = {t: cpu(t) for t in tids()}; t0 = time.perf_counter(); fn(); el = time.perf_counter() - t0
    rows = sorted(((cpu(t) - c0.get(t, 0.0), t) for t in tids()), reverse=True)[:6]
    print("%s (%.2f s wall): " % (title, el) + ", ".join("tid %s %.0f%%" % (t, 100 * d / el) for d, t in rows if d > 0.005), flush=True)
report("idle host, idle GPU (sleep 1 s)", lambda: time.sleep(1.0))
def busy():
    for _ in range(25): step()
    torch.cuda.synchronize()
report("25 training steps", busy)
def gpu_only():
    a = torch.randn(8192, 8192, device=dev)
    for _ in range(60): a = a @ a * 1e-4
    time.sleep(0.5); torch.cuda.synchronize()
report("GPU busy with plain GEMMs, host asleep", gpu_only)
print("main tid", os.getpid())
for k in ("HSA_ENABLE_INTERRUPT", "GPU_MAX_HW_QUEUES", "AMD_DIRECT_DISPATCH", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_IPC_MODE_LEGACY"):
    print(k, "=", os.environ.get(k))
pf.close()
