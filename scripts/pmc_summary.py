"""Per-kernel summary of the rocprofv3 --pmc passes (rocpd sqlite, view counters_collection):
HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM), the algorithmic bytes per launch the
bench line of the same tree reports (-> traffic_ratio) and MFMA / VALU busy fractions; stamped with the hash of the kernel
sources so that bench.py only attaches it to runs of the same kernels.
usage: pmc_summary.py <dir with pmc_final_*/p_results.db> <out.json> [<bench json line with roofline.per_kernel>]"""
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

def demangled_short(name):
    # rocpd stores demangled names: "void conv_fwd_kernel<2, 2, 1, 2, 0, true, true>(...)"
    if 'lstm_step_fwd_bf' in name:
        return 'lstm_step_fwd_bf16x6<64x32>'
    if 'lstm_step_fwd' in name:       # lstm_step_fwd_kernel and lstm_step_fwd_k2_kernel, with or without a leading "void "
        return 'lstm_step_fwd<64x64>'
    if 'conv_bfh_kernel' in name:         # the 128 x 128 tile on 16-k stages (round 6): <WM, WN, ...> waves of 64 x 64
        mh = re.match(r'(?:void )?conv_bfh_kernel<([^>]*)>', name)
        wm, wn = [int(a) for a in mh.group(1).split(',')[:2]] if mh else (2, 2)
        return 'conv_bf16x6<%dx%d>' % (wm * 64, wn * 64)
    mb = re.match(r'(?:void )?conv_bf_kernel<([^>]*)>', name)
    if mb:
        wm, wn, sm, sn = [int(a) for a in mb.group(1).split(',')[:4]]
        return 'conv_bf16x6<%dx%d>' % (wm * sm * 32, wn * sn * 32)
    if 'conv_wgrad128_bf_kernel' in name:
        return 'conv_wgrad128_bf16x6<128x128>'
    if 'conv_wgrad128_kernel' in name:
        return 'conv_wgrad128<128x128>'
    m = re.match(r'void (conv_fwd_kernel|conv_ut_kernel|conv_wgrad_kernel|narrow_fwd_kernel)<([^>]*)>', name)
    if not m:
        return None
    k, args = m.group(1), [a.strip() for a in m.group(2).split(',')]
    if k in ('conv_fwd_kernel', 'conv_ut_kernel'):      # same tile configurations, same report name
        wm, wn, sm, sn, bm = [int(a) for a in args[:5]]
        return 'conv_fwd<%dx%d,%s>' % (wm * sm * 32, wn * sn * 32, 'NK' if bm else 'KN')
    if k == 'conv_wgrad_kernel':
        wm, wn, sm, sn = [int(a) for a in args[:4]]
        return 'conv_wgrad<%dx%d>' % (wm * sm * 32, wn * sn * 32)
    return 'narrow_fwd<%s>' % ('transposed' if int(args[0]) == 1 else 'conv')


def mangled_short(name):
    """The same report names from the MANGLED symbol (the kernel_symbol table of a kernel trace keeps `_Z14conv_ut_kernelILi2E...`)."""
    m = re.match(r'_Z\d+(conv_fwd_kernel|conv_ut_kernel|conv_bfh_kernel|conv_bf_kernel|conv_wgrad_kernel|narrow_fwd_kernel|narrow_sc_kernel|'
                 r'conv_wgrad128_bf_kernel|conv_wgrad128_kernel|lstm_step_fwd_bf_kernel|lstm_step_fwd_kernel|lstm_step_fwd_k2_kernel)(?:I((?:L[ib]\d+E)+)E)?', name)
    if not m:
        return None
    k = m.group(1)
    args = [int(a) for a in re.findall(r'L[ib](\d+)E', m.group(2) or '')]
    if k == 'lstm_step_fwd_bf_kernel':
        return 'lstm_step_fwd_bf16x6<64x32>'
    if k.startswith('lstm_step_fwd'):
        return 'lstm_step_fwd<64x64>'
    if k == 'conv_wgrad128_bf_kernel':
        return 'conv_wgrad128_bf16x6<128x128>'
    if k == 'conv_wgrad128_kernel':
        return 'conv_wgrad128<128x128>'
    if k == 'conv_bfh_kernel':
        return 'conv_bf16x6<%dx%d>' % (args[0] * 64, args[1] * 64)
    if k == 'conv_bf_kernel':
        wm, wn, sm, sn = args[:4]
        return 'conv_bf16x6<%dx%d>' % (wm * sm * 32, wn * sn * 32)
    if k in ('conv_fwd_kernel', 'conv_ut_kernel'):
        wm, wn, sm, sn, bm = args[:5]
        return 'conv_fwd<%dx%d,%s>' % (wm * sm * 32, wn * sn * 32, 'NK' if bm else 'KN')
    if k == 'conv_wgrad_kernel':
        wm, wn, sm, sn = args[:4]
        return 'conv_wgrad<%dx%d>' % (wm * sm * 32, wn * sn * 32)
    return 'narrow_fwd<%s>' % ('transposed' if args[0] == 1 else 'conv')


def collect(db):
    c = sqlite3.connect(db)
    out = defaultdict(lambda: defaultdict(list))
    for name, counter, value in c.execute('select kernel_name, counter_name, value from counters_collection'):
        s = demangled_short(name)
        if s:
            out[s][counter].append(float(value))
    return out


def main(base, dst, bench_json=None):
    import os as _os
    import sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    import bench as _bench
    alg, pk = {}, {}
    if bench_json and _os.path.exists(bench_json):
        lines = [l for l in open(bench_json).read().splitlines() if l.startswith('{')]
        if lines:
            pk = json.loads(lines[-1]).get('roofline', {}).get('per_kernel', {})
            alg = {k: v.get('algorithmic_bytes_per_launch') for k, v in pk.items()}
    res = {'source': 'rocprofv3 --kernel-trace --pmc <one pass each> of `python bench.py --steps 2 --warmup 1 --no-graphs '
                     '--no-kernel-events` (scripts/run_profile_pmc.sh), 1x MI355X, batch 32, 192x192',
           'csrc_hash': _bench._csrc_hash(),
           'correction': 'FETCH_SIZE (KB) x2 (gfx950 tallies the 128-B requests of coalesced 16-B/lane reads at 64 B, '
                         'MI355X_MICROARCH.md section HBM); WRITE_SIZE (KB) as reported',
           'kernels': {}}
    fetch = collect(os.path.join(base, 'pmc_final_FETCH_SIZE', 'p_results.db'))
    write = collect(os.path.join(base, 'pmc_final_WRITE_SIZE', 'p_results.db'))
    sq = collect(os.path.join(base, 'pmc_final_SQ_VALU_MFMA_BUSY_CYCLES', 'p_results.db'))
    for k in sorted(set(fetch) | set(sq)):
        e = {}
        if k in fetch and k in write:
            f = sum(fetch[k]['FETCH_SIZE']) / len(fetch[k]['FETCH_SIZE']) * 1024
            w = sum(write[k]['WRITE_SIZE']) / len(write[k]['WRITE_SIZE']) * 1024
            e.update(launches_profiled=len(fetch[k]['FETCH_SIZE']), fetch_bytes_per_launch_raw=f,
                     fetch_bytes_per_launch_corrected=2 * f, write_bytes_per_launch=w, hbm_bytes_per_launch=2 * f + w)
            # launch by launch (dispatch order is the same in every pass: one deterministic command): which layers re-read
            if len(fetch[k]['FETCH_SIZE']) == len(write[k]['WRITE_SIZE']) and len(fetch[k]['FETCH_SIZE']) <= 200:
                e['hbm_mb_by_launch'] = [round((2 * a + b) * 1024 / 1e6, 1) for a, b in zip(fetch[k]['FETCH_SIZE'], write[k]['WRITE_SIZE'])]
            if alg.get(k):
                e.update(algorithmic_bytes_per_launch=alg[k], traffic_ratio=(2 * f + w) / alg[k])
        if k in sq and sq[k].get('GRBM_GUI_ACTIVE'):
            tot = lambda n: sum(sq[k].get(n, [0.0]))
            gui = tot('GRBM_GUI_ACTIVE') / 8.0      # the counter is summed over the 8 XCDs: /8 = shader-clock cycles
            # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs (256 CUs x 4) and equals 64 x #MFMA for 32x32x2 f32;
            # SQ_ACTIVE_INST_VALU counts quad-cycles
            simd_cycles = gui * 1024
            e.update(mfma_busy_frac=tot('SQ_VALU_MFMA_BUSY_CYCLES') / simd_cycles if gui else None,
                     valu_busy_frac=tot('SQ_ACTIVE_INST_VALU') * 4 / simd_cycles if gui else None,
                     # busy cycles per instruction: 64 for v_mfma_f32_32x32x2_f32, 32 for v_mfma_f32_32x32x16_bf16
                     mfma_instructions=tot('SQ_VALU_MFMA_BUSY_CYCLES') / (32.0 if 'bf16x6' in k else 64.0),
                     valu_instructions=tot('SQ_INSTS_VALU'),
                     kernel_cycles=gui, launches_profiled_sq=len(sq[k]['GRBM_GUI_ACTIVE']))
        res['kernels'][k] = e
    # the same kernels INSIDE the replayed step: a kernel trace of the graph-replayed bench (every iteration launches the same
    # kernels: iterations = calls / launches per step of the bench line) -> time per iteration and the rate on the bench line's
    # FLOPs.  A launch shares the chip with the other chains of the step there: this is the figure that bounds the step.
    tdb = os.path.join(base, 'trace_replay', 'p_results.db')
    if os.path.exists(tdb) and pk:
        c = sqlite3.connect(tdb)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        ks = [t for t in tabs if 'kernel_symbol' in t][0]
        kd = [t for t in tabs if 'kernel_dispatch' in t][0]
        agg = defaultdict(lambda: [0.0, 0])
        for name, s0, e0 in c.execute('select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id' % (kd, ks)):
            sh = demangled_short(name) or mangled_short(name)
            if sh:
                agg[sh][0] += (e0 - s0) / 1e3
                agg[sh][1] += 1
        for k, (us, calls) in agg.items():
            L = pk.get(k, {}).get('launches_per_step')
            if not L or k not in res['kernels']:
                continue
            iters = calls / L
            ms = us / iters / 1e3
            res['kernels'][k].update(in_step_ms_per_step=ms, in_step_launches_traced=calls,
                                     in_step_tflops=pk[k]['flop_per_launch'] * L / (ms * 1e-3) / 1e12)
    json.dump(res, open(dst, 'w'), indent=1)
    for k, e in res['kernels'].items():
        print('%-26s hbm/launch %8.1f MB  mfma_busy %s  valu_busy %s' % (
            k, e.get('hbm_bytes_per_launch', 0) / 1e6, ('%.3f' % e['mfma_busy_frac']) if e.get('mfma_busy_frac') else '-',
            ('%.3f' % e['valu_busy_frac']) if e.get('valu_busy_frac') else '-'))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
