"""What the process's first HIP calls cost, with and without libspacecarve's code objects registered."""
import ctypes, os, sys, time
which = sys.argv[1]
t0 = time.perf_counter()
if which == "torchhip":
    import importlib.util
    spec = importlib.util.find_spec("torch")
    hip = ctypes.CDLL(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
elif which == "rocmhip":
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
else:
    sys.path.insert(0, "/root/repo")
    os.environ["SC_PREWARM"] = "0"
    from plant3dvision_amd import _native as nat
    nat.backend()
    hip = nat.hip_runtime()
t1 = time.perf_counter()
hip.hipInit(0)
t2 = time.perf_counter()
hip.hipSetDevice(0)
hip.hipFree(ctypes.c_void_p(0))
t3 = time.perf_counter()
s = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)
t4 = time.perf_counter()
s2 = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(s2), 1)
t5 = time.perf_counter()
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20)
hip.hipMemsetAsync(p, 0, 1 << 20, s)
hip.hipStreamSynchronize(s)
t6 = time.perf_counter()
print(which, "load %.1f init %.1f setdevice+free0 %.1f stream1 %.1f stream2 %.1f first memset %.1f ms" % tuple((b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))), flush=True)
