import ctypes, sys, numpy as np, torch
W = ctypes.CDLL(sys.argv[2]); L = ctypes.CDLL(sys.argv[1])
def run(lib, fn, x):
    o = np.zeros_like(x); getattr(lib, fn)(x.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size)); return o
rng = np.random.default_rng(0)
def cmp(a,b): return int(((a.view(np.uint32)!=b.view(np.uint32)) & ~(np.isnan(a)&np.isnan(b))).sum())
for name, mine, s16, s8, tf, x in (("exp","sl_exp_arr","sw_exp16","sw_exp8",torch.exp, rng.uniform(-12,12,1<<21).astype(np.float32)),
                                   ("tanh","sl_tanh_arr","sw_tanh16","sw_tanh8",torch.tanh, (rng.standard_normal(1<<21)*1.5).astype(np.float32))):
    m = run(L, mine, x); a16 = run(W, s16, x); a8 = run(W, s8, x); t = tf(torch.from_numpy(x)).numpy()
    print(name, 'mine vs sleef16', cmp(m,a16), 'sleef16 vs sleef8', cmp(a16,a8), 'torch vs sleef16', cmp(t,a16), 'torch vs mine', cmp(t,m))
