import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    r=d["roofline"]
    print(f.split('/')[-1], "value",round(d["value"]), "us/step",round(d["ms_per_step"]*1e3,2), "step",round(r["step"],3), "frac",round(r["frac"],3),"alone_us",round(r["launch_ms_alone"]*1e3,2),"frac_alone",round(r["frac_alone"],3), r["bound"], r["kernel"], "traffic", r["traffic"] and round(r["traffic"]/1e6,1))
