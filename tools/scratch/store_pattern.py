import ctypes as C, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libstore_pattern.so")
if not os.path.exists(SO):  # throw-away tool: built on demand
    os.system(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o {SO} {os.path.join(HERE, 'store_pattern.hip')}")
lib = C.CDLL(SO)
lib.run_store.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = lambda: torch.cuda.current_stream().cuda_stream
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for M, N in ((2047872, 512), (31872, 480), (63872, 768), (2047872, 256)):
    x = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    r = [t(lambda m=m: lib.run_store(x.data_ptr(), M, N, m, st())) for m in (0, 1, 2)]
    print(f"M={M} N={N}: " + "  ".join(f"mode{m} {s*1e6:8.1f} us {M*N*2/s/1e12:5.2f} TB/s" for m, s in enumerate(r)))
