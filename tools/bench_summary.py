import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value %.1f"%d["value"], "steady", json.dumps(d.get("steady_state"))[:200])
    except Exception as e:
        print(f, "ERR", e)
