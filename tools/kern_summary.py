import json,sys
for f in sys.argv[1:]:
    try:
        line=[l for l in open(f) if l.startswith('{')][-1]
    except Exception as e:
        print(f, "no json"); continue
    d=json.loads(line)
    print(f, "value=%.2f img/s ms/step=%.1f"%(d['value'], d['ms_per_step']))
    for k in d['kernels']:
        print("   %-62s n=%5d ms=%8.2f tf=%s"%(k['name'][6:],k['launches'],k['ms'],k['tflops']))
