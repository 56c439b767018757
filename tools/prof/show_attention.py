import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["shape"][:44], "one_pass", d.get("one_pass_us"), "resident", d.get("one_pass_resident_us"), "kw8", d.get("one_pass_resident_8_key_waves_us"), "err", d.get("one_pass_vs_steps_max_rel"))
