import json, sys, re
for path in sys.argv[1:]:
    line = [l for l in open(path) if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    eng = d["config"]["engine"]
    print(path, "C2 %.1f fps %.4f ms | class_map %s fps" % (d["value"], d["ms_per_step"], d.get("class_map", {}).get("value")))
    print("  ", eng[eng.index("ms/frame"):][:700])
    print("   parity", d["parity"]["pass"], d["parity"]["rel_to_max_logit"], d["parity"]["argmax_agreement"])
