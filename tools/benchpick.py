import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f).read().splitlines() if x.startswith("{")][-1]
        d=json.loads(l)
        print(f, "value", d["value"], "ms/step", d["ms_per_step"], "stages", d.get("stages_ms"), "decomp", d.get("decompress",{}).get("value"), "phrase", d.get("phrase_book_variant",{}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
