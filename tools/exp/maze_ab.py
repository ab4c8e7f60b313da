import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from unreal_amd import ops, _lib
B = 4096
ring = ops.Ring(B, 8, "cuda:0")
ops.maze_reset(ring)
acts = torch.randint(0, 4, (B,), dtype=torch.int32, device="cuda:0")
P = ctypes.c_void_p
def run(lib):
    pt = lambda t: P(t.data_ptr())
    lib.unreal_maze_step(B, ring.H1, pt(acts), None, pt(ring.pos), pt(ring.last_action), pt(ring.last_reward), pt(ring.count),
                         pt(ring.frames), pt(ring.r_reward), pt(ring.r_action), pt(ring.r_terminal), pt(ring.r_last_action),
                         pt(ring.r_last_reward), pt(ring.r_pc), None, None, None, None, None, 1, 0,
                         P(torch.cuda.current_stream().cuda_stream))
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for n in (16, 8, 4, 2):
    path = os.path.join(root, "tools/exp/build/libenv_%d.so" % n) if n != 16 else None
    if n == 16:
        continue
    lib = ctypes.CDLL(path)
    for _ in range(5): run(lib)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run(lib)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("actors per workgroup %2d: %.1f us  %.0f GB/s" % (n, ms * 1e3, B * 22768 / ms / 1e6))
