import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d.get("paths"),indent=1)[:4000]); print(json.dumps(d["configs"].get("cfg1"),indent=1)); print(d["configs"]["cfg4"]["round_kernels"]["fold_round_evals_kernel"]); print(d["configs"]["cfg4"]["round_kernels"]["round_evals_kernel"]); print(d["configs"]["cfg2"]["absorb_GBps"], d["configs"]["cfg4"]["absorb_GBps"], d["prewarm_s"], d["value"], d["roofline"]["frac"], d["configs"]["cfg4"]["device_ms_per_layer"])
