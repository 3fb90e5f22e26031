"""Per-kernel PMC counters of one command, one rocprofv3 pass per counter set (--pmc only with --kernel-trace: the pool's rule).

python tools/pmc_kernels.py out.json "SET1 counters..." ["SET2 ..." ...] -- <command ...>

Sums every counter over the dispatches of each kernel (name cut at the template / argument list) and divides by the dispatch count;
writes {kernel: {counter: per-dispatch value, "dispatches": n}}."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys


def main():
    argv = sys.argv[1:]
    sep = argv.index("--")
    out_path, sets, cmd = argv[0], argv[1:sep], argv[sep + 1:]
    res = collections.defaultdict(dict)
    for i, cs in enumerate(sets):
        d = "/tmp/pmck_%d" % os.getpid()
        shutil.rmtree(d, ignore_errors=True)
        full = ["rocprofv3", "--kernel-trace", "--pmc"] + cs.split() + ["-d", d, "-o", "out", "--output-format", "csv", "--"] + cmd
        r = subprocess.run(full, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=int(os.environ.get("PMC_TIMEOUT", "240")))
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not f:
            print("pass %d (%s): no counter file, rc %d: %s" % (i + 1, cs, r.returncode, r.stdout.decode(errors="replace")[-300:]), flush=True)
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for row in csv.DictReader(open(f[0])):
            k = row["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").strip()
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row.get("Dispatch_Id"))
        for k in acc:
            nd = max(1, len(disp[k]))
            res[k]["dispatches"] = nd
            for c, v in acc[k].items():
                res[k][c] = v / nd
        print("pass %d (%s): %d kernels" % (i + 1, cs, len(acc)), flush=True)
        shutil.rmtree(d, ignore_errors=True)
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print(k, json.dumps(res[k], sort_keys=True))


if __name__ == "__main__":
    main()
