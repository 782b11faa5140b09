import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    k=d["kernel_families"]
    def g(n): return k.get(n,{}).get("ms_per_step")
    print(f.split("/")[-1], d["value"], d["ms_per_step"], "fwd",g("conv3x3_fwd_bf16"),"dgrad",g("conv3x3_dgrad_bf16"),"wgrad",g("conv3x3_wgrad_bf16"),"red",g("wgrad_reduce"),"tconv",g("tconv2x2_bf16"),"tw",g("tconv2x2_wgrad_bf16"), "frac", d["roofline"]["frac"])
