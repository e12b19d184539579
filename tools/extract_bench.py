import json, sys
for path in sys.argv[1:]:
    try:
        line = [l for l in open(path) if l.startswith('{"metric"')][-1]
        d = json.loads(line)
        w = d["workloads"]
        print(path, "| C2 %.1f fps" % d["value"], "|", " | ".join("%s %.2f ms" % (k.split("_")[0], v["ms_per_step"]) for k, v in w.items() if k != "C2_student_infer"))
    except Exception as e:
        print(path, "ERR", e, open(path).read()[-600:])
