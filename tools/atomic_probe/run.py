import ctypes, os, torch, json
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "probe.so"))
lib.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
heads, S, kb = 24, 2432, 10
acc = torch.zeros(heads, S, 128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
out = {}
for mode, name in ((0, "unsafeAtomicAdd"), (1, "atomicAdd"), (2, "workgroup-scope fetch_add")):
    for xa in (0, 1):
        for _ in range(3): lib.run(acc.data_ptr(), heads, S, kb, mode, xa, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.run(acc.data_ptr(), heads, S, kb, mode, xa, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        nbytes = heads * kb * (S // 64) * 64 * 128 * 4
        out[f"{name} xcd_aware={xa}"] = {"us": round(us, 1), "atomic_GBps": round(nbytes / us / 1e3, 0), "MB": round(nbytes / 1e6)}
acc.zero_(); lib.run(acc.data_ptr(), heads, S, kb, 0, 1, st); torch.cuda.synchronize()
exp = kb * sum(1.0 + i for i in range(32)) / 32   # each element gets kb adds of (1+i) for its own i
print(json.dumps(out, indent=1)); print("sum check", acc.sum().item(), heads * S * 128 * kb * (sum(1.0 + i for i in range(32)) / 32))
