"""The recurrent step alone: full launch vs. its epilogue only (with_gemm = 0), bf16x6 vs exact fp32, at the caption branch's two
row counts.  usage: lstm_step_probe.py [rows ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import torch
from sketchyscenecolorization_amd import hip

C = int(os.environ.get("LSTM_C", "512"))
for rows in [int(a) for a in sys.argv[1:]] or [576, 16, 1152]:
    g = torch.Generator(device='cuda').manual_seed(1)
    h = torch.randn(rows, C, device='cuda', generator=g) * 0.5
    c = torch.randn(rows, C, device='cuda', generator=g) * 0.5
    K = torch.randn(C, 4 * C, device='cuda', generator=g) * 0.04
    hip.register_param_buffer(K)        # persistent planes, as a parameter has them
    g1 = torch.randn(rows, 4 * C, device='cuda', generator=g)
    g2 = torch.randn(max(rows // 36, 1), 4 * C, device='cuda', generator=g)
    div2 = 36 if rows % 36 == 0 else rows
    mask = torch.ones(max(rows // div2, 1), dtype=torch.int32, device='cuda')
    co, ho = torch.empty_like(c), torch.empty_like(h)
    acts = torch.empty(rows, 4 * C, device='cuda')
    hp = torch.zeros(2, hip.lstm_hplanes_floats(rows, C), device='cuda')
    hip.call('ssc_lstm_hsplit', h, rows, C, hp[0])
    for label, kw, wg in (('bf16x6 full', dict(hp_in=hp[0], hp_out=hp[1]), True), ('bf16x6 epilogue only', dict(hp_out=hp[1]), False),
                          ('bf16x6 full, no acts', dict(hp_in=hp[0], hp_out=hp[1], no_acts=True), True),
                          ('fp32 full', dict(exact=True), True), ('fp32 epilogue only', dict(exact=True), False),
                          ('unfused: conv kernel GEMM + gate kernel', None, True)):
        tmp = torch.empty(rows, 4 * C, device='cuda')

        def run():
            if kw is None:      # the two-launch form of text_fusion.py (SSC_LSTM_FUSED=0)
                hip.matmul(h, K, tmp)
                hip.call('ssc_lstm_pointwise_fwd', tmp, g1, g2, div2, mask, div2, c, h, rows, C, co, ho, acts)
                return
            k2 = dict(kw)
            a = None if k2.pop('no_acts', False) else acts
            hip.lstm_step_fwd(h, K, 4 * C, g1, g2, div2, mask, div2, c, rows, C, wg, co, ho, a, **k2)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(50):
                    run()
            gr.replay()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                gr.replay()
            e1.record()
            st.synchronize()
        print('rows %5d  %-28s %7.2f us' % (rows, label, e0.elapsed_time(e1) * 1000 / 200))
