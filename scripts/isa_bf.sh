# ISA of the hot conv_bf_kernel instantiations (igemm_bf16.hip): main-loop instruction mix, registers
set -e
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Wno-unused-value -Wno-unused-function -DSSC_ISA_ONLY $SSC_EXTRA_FLAGS -S --cuda-device-only \
  sketchyscenecolorization_amd/csrc/igemm_bf16.hip -o /tmp/isa/bf.s 2>/dev/null
python3 - <<'PY'
import re
from collections import Counter
lines=open('/tmp/isa/bf.s').read().split('\n')
starts=[i for i,l in enumerate(lines) if re.match(r'^_Z\d+conv_\w+_kernel.*:\s', l)]
for st in starts:
    end=st
    while 's_endpgm' not in lines[end]: end+=1
    fn=[l for l in lines[st:end] if l.strip() and not l.strip().startswith(';')]
    mf=[i for i,l in enumerate(fn) if 'v_mfma' in l]
    lo=mf[0]
    while not fn[lo].startswith('.LBB'): lo-=1
    hi=mf[-1]
    while 's_cbranch' not in fn[hi]: hi+=1
    body=[l for l in fn[lo:hi+1] if not l.startswith('.')]
    c=Counter(l.split()[0] for l in body)
    valu=sum(v for k,v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    print(lines[st].split(':')[0][:60], ' loop instrs', len(body), ' mfma', sum(v for k,v in c.items() if k.startswith('v_mfma')), ' other VALU', valu,
          ' ds', sum(v for k,v in c.items() if k.startswith('ds_')), ' global', sum(v for k,v in c.items() if k.startswith('global_')),
          ' salu', sum(v for k,v in c.items() if k.startswith('s_')))
    print('   ', sorted(((k,v) for k,v in c.items() if k.startswith('v_') and not k.startswith('v_mfma')), key=lambda kv:-kv[1]))
    print('   ', sorted(((k,v) for k,v in c.items() if k.startswith('ds_') or k.startswith('global') or k.startswith('s_')), key=lambda kv:-kv[1]))
print([l.strip() for l in lines if '.vgpr_count:' in l or '.agpr_count:' in l or '.private_segment_fixed_size:' in l or 'vgpr_spill' in l])
PY
