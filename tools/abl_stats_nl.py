"""Timing-only ablations of attn_stats_nl_tri (tools/scratch/nlabl/lib_<mask>.so = the library built with
-DSAMBLE_NL_ABL=<mask>: 8 the compiler's own operand-read placement, 16 half the K operand reads (every pair of k-steps
shares one read: wrong logits), 2 no matrix products, 32 no neighbour-logit extraction, 64 no exponentials) against the
shipped library, on the metric shapes with real
images.  Build: see the loop at the end of tools/build_scratch_libs.sh; run on the GPU box."""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import ops, synth, _lib
B, N, nt, K = 32, 2048, 6, 32
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.normal((B, 128, N), 1)).to(dev)
tokens = torch.from_numpy(synth.normal((128, nt), 2)).to(dev)
w = (torch.from_numpy(synth.normal((384, 128), 3)) * 0.1).to(dev)
qkv, imgs = ops.stage_proj_fwd(x, tokens, w, images="fwd")
nn_idx = ops.stage_knn(x, x, K)
nn_sorted, masks = ops.stage_nn_prepare(nn_idx)
lse = torch.empty((B, N), device=dev)
tok = torch.empty((B, N, nt), device=dev)
ws = ops.score_workspace(B, N, 6, dev)
here = os.path.dirname(os.path.abspath(__file__))
libs = [("shipped", _lib.LIB_PATH)] + [(os.path.basename(f), f) for f in sorted(glob.glob(os.path.join(here, "scratch", "nlabl", "lib_*.so")))]
for rep in range(2):
    for name, f in libs:
        lib = ctypes.CDLL(f)
        fn = lib.samble_attn_stats_nl_tri_f32
        fn.argtypes = _lib._SIGNATURES["samble_attn_stats_nl_tri_f32"][1]
        st = torch.cuda.current_stream().cuda_stream
        def run():
            rc = fn(imgs[0].data_ptr(), imgs[1].data_ptr(), B, N, nt, 128, masks.data_ptr(), K, None, lse.data_ptr(),
                    tok.data_ptr(), nn_sorted.data_ptr(), 2, ws.data_ptr(), ws.numel(), 0, st)
            assert rc == 0, rc
        for _ in range(5): run()
        torch.cuda.synchronize()
        if name == "shipped":
            ref = (lse.clone(), tok.clone(), ws.clone())
        elif rep == 0:
            same = [bool(torch.equal(a, b2)) for a, b2 in zip(ref, (lse, tok, ws))]
            print(f"{name}: lse / token logits / score workspace equal to the shipped library's: {same}")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:12s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
