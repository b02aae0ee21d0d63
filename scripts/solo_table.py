"""Table of a scripts/solo_sweep_all.sh output directory: one row per configuration, one column per case."""
import glob, json, os, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl")))
head = None
for f in files:
    d = [json.loads(l) for l in open(f) if l.startswith("{")]
    if not d:
        print(os.path.basename(f), "EMPTY", open(f).read()[-300:]); continue
    if head is None:
        head = [x["case"] for x in d]
        print("%-20s" % "", *["%12s" % c[:12] for c in head])
    print("%-20s" % os.path.basename(f)[:-6], *["%12.3f" % x["ms_median"] for x in d], "" if all(x["same_image"] for x in d) else "IMAGE DIFFERS")
