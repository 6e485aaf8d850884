"""Scratch first-light check on the GPU box: ABC + HBV vs oracle, timing."""
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from rrmpg_amd.utils import synthetic as syn
import torch

lib = ctypes.CDLL("rrmpg_amd/librrhip.so")
lib.rr_last_error.restype = ctypes.c_char_p
f64p = ctypes.POINTER(ctypes.c_double); i8p = ctypes.POINTER(ctypes.c_int8)
i64 = ctypes.c_int64; dbl = ctypes.c_double; vp = ctypes.c_void_p
lib.rr_hbvedu_simulate.argtypes = [f64p, f64p, i8p, f64p, f64p, i64] + [dbl]*4 + [f64p, i64] + [f64p]*7
lib.rr_abc_simulate.argtypes = [f64p, i64, dbl, f64p, i64, f64p, f64p, f64p, f64p]
lib.rr_hbvedu_simulate_dev.argtypes = [vp]*5 + [i64] + [dbl]*4 + [vp, i64] + [vp]*5 + [i64, vp, vp, vp, ctypes.c_size_t, vp]
lib.rr_hbvedu_workspace_bytes.restype = ctypes.c_size_t
lib.rr_hbvedu_workspace_bytes.argtypes = [i64, i64]
lib.rr_abc_simulate_dev.argtypes = [vp, i64, dbl, vp, i64, vp, vp, i64, vp, vp, vp, ctypes.c_size_t, vp]
P = lambda a: a.ctypes.data_as(f64p)
print("devices", lib.rr_device_count())
f = syn.make_forcing(syn.T_30YR)
T = syn.T_30YR
rng = np.random.default_rng(0)
lo = np.array([-1,3,100,1,0.01,90,0.05,0.01,0.01,0.01,2.]); hi = np.array([1,7,200,7,0.07,180,0.2,0.1,0.05,0.05,5.])
N = 1000
par = lo + (hi-lo)*rng.random((N, 11))
m0 = (f["month"]-1).astype(np.int8)
ref = pyoracle.simulate_hbvedu(f["temp"], f["prec"], m0, f["PE_m"], f["T_m"], (0,100,3,10), par, return_storage=True, nthreads=8)
outs = [np.zeros((T, N)) for _ in range(5)]
qobs = ref[0][:, 0].copy(); sse = np.zeros(N)
rc = lib.rr_hbvedu_simulate(P(f["temp"]), P(f["prec"]), m0.ctypes.data_as(i8p), P(f["PE_m"]), P(f["T_m"]), T, 0., 100., 3., 10., P(par), N, *[P(o) for o in outs], P(qobs), P(sse))
print("rc", rc, lib.rr_last_error())
for o, r, n in zip(outs, ref, ["q","snow","soil","s1","s2"]):
    err = np.max(np.abs(o-r)/np.maximum(np.abs(r), 1e-9))
    print("hbv", n, "max rel err", err, "bit-equal frac", np.mean(o == r))
sse_ref = ((qobs[:, None]-ref[0])**2).sum(0)
print("sse rel", np.max(np.abs(sse-sse_ref)/np.maximum(sse_ref,1e-9)))
# ABC
pa = rng.random((N, 3))*np.array([1,0.3,1.])
r = pyoracle.simulate_abc(f["prec"], 2.0, pa, return_storage=True)
oq = np.zeros((T,N)); os_ = np.zeros((T,N))
rc = lib.rr_abc_simulate(P(f["prec"]), T, 2.0, P(pa), N, P(oq), P(os_), None, None)
print("abc rc", rc, "bit-exact", np.array_equal(oq, r[0]), np.array_equal(os_, r[1]))
# odd N
oq = np.zeros((T,N-1)); rc = lib.rr_abc_simulate(P(f["prec"]), T, 2.0, P(pa), N-1, P(oq), None, None, None)
print("abc odd N rc", rc, np.array_equal(oq, r[0][:, :N-1]))

# device-resident timing
dev = torch.device("cuda:0")
def tt(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for N in (100_000, 1_000_000):
    par = lo + (hi-lo)*rng.random((N, 11))
    d = dict(temp=tt(f["temp"]), prec=tt(f["prec"]), month=tt(m0), pe=tt(f["PE_m"]), tm=tt(f["T_m"]), par=tt(par))
    q = torch.empty((T, N), dtype=torch.float64, device=dev)
    wsb = lib.rr_hbvedu_workspace_bytes(T, N); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run(qq):
        rc = lib.rr_hbvedu_simulate_dev(d["temp"].data_ptr(), d["prec"].data_ptr(), d["month"].data_ptr(), d["pe"].data_ptr(), d["tm"].data_ptr(), T, 0.,100.,3.,10., d["par"].data_ptr(), N, qq, None,None,None,None, N, None, None, ws.data_ptr(), wsb, st)
        assert rc == 0, lib.rr_last_error()
    for mode, qq in (("qsim", q.data_ptr()),):
        run(qq); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): run(qq)
        torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/3
        print(f"HBV N={N} {mode}: {dt*1e3:.2f} ms  {N*T/dt:.3e} steps/s  {8*N*T/dt/1e12:.3f} TB/s")
    # ABC
    pa = tt(rng.random((N, 3))*np.array([1,0.3,1.]))
    def runa():
        rc = lib.rr_abc_simulate_dev(d["prec"].data_ptr(), T, 2.0, pa.data_ptr(), N, q.data_ptr(), None, N, None, None, ws.data_ptr(), wsb, st)
        assert rc == 0
    runa(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): runa()
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/3
    print(f"ABC N={N} qsim: {dt*1e3:.2f} ms  {N*T/dt:.3e} steps/s  {8*N*T/dt/1e12:.3f} TB/s")
    del q
