import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('img/s %.1f  ms %.3f  all_igemm %.1f TF  exec frac %.3f' % (p["value"], p["ms_per_step"], p["roofline"]["all_igemm_tflops"], p.get("step_frac_of_fp32_peak",0)))
for k,v in p["roofline"]["per_kernel"].items(): print("   %-24s %6.1f TF %7.3f ms  %5.1f" % (k,v["tflops"],v["ms_per_step"],v["launches_per_step"]))
if 'generator_fwd_bwd' in p: print('gen fwd+bwd', p['generator_fwd_bwd']['ms'], p['generator_fwd_bwd']['frac_of_fp32_mfma_peak_executed'])
