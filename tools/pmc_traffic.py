"""HBM-side traffic per kernel launch from rocprofv3 PMC counters (GPU box): two separate --pmc passes (FETCH_SIZE,
WRITE_SIZE — they do not fit one pass, MI355X_MICROARCH.md) of `bench.py --steps 2 --warmup 1 --no-cpu --no-conv`,
per-dispatch median, gfx950 correction FETCH_SIZE x2 (calibrated on the streaming k_cg_update in round 1).
Writes a text digest and the JSON bench.py reads for roofline.traffic.

usage: python tools/pmc_traffic.py <out_dir> <tag> [workload]      -> <out_dir>/<tag>_pmc_digest.txt, <out_dir>/<tag>_pmc_traffic.json
Never combines --pmc with a trace domain other than --kernel-trace."""
import csv, glob, json, os, re, statistics, subprocess, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out_dir, tag = os.path.abspath(sys.argv[1]), sys.argv[2]          # rocprofv3 runs from /tmp: relative paths would land there
os.makedirs(out_dir, exist_ok=True)
workload = sys.argv[3] if len(sys.argv) > 3 else "c3_100k"
cmd = ["python3", os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-conv", "--workload", workload]
os.environ.setdefault("TMPDIR", "/tmp")


def short(name):
    m = re.match(r"(?:void )?(?:tsgo::)?([A-Za-z_0-9]+(?:<[^(]*>)?)", name)
    return m.group(1) if m else name


def run(counter):
    d = os.path.join(out_dir, "pmc_" + counter.lower())
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", counter.lower(), "--"] + cmd,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return per


fetch, write = run("FETCH_SIZE"), run("WRITE_SIZE")
lines = ["rocprofv3 --pmc passes (separate runs) of: " + " ".join(cmd[:1] + ["bench.py"] + cmd[2:]) + "   [%s, f64]" % workload,
         "Per dispatch, KB as reported (median over the dispatches that did work: converged PCG iterations exit early).",
         "gfx950 correction: FETCH_SIZE x2 for coalesced streaming reads, WRITE_SIZE exact.", ""]
kern = {}
for k in sorted(set(fetch) | set(write)):
    def med(v):
        live = [x for x in v if x > 0.05 * max(v)] if v else []
        return (statistics.median(live) if live else 0.0), len(v), len(live)
    f, nf, lf = med(fetch.get(k, [])); w, nw, lw = med(write.get(k, []))

    def mean_live(v):      # one symbol may run on several multigrid levels: the MEAN over its live dispatches is what a per-launch average needs
        live = [x for x in v if x > 0.05 * max(v)] if v else []
        return sum(live) / len(live) if live else 0.0
    kern[k] = {"fetch_kb_raw": f, "write_kb": w, "hbm_bytes_corrected": (2 * f + w) * 1024.0,
               "hbm_bytes_mean": (2 * mean_live(fetch.get(k, [])) + mean_live(write.get(k, []))) * 1024.0}
    lines.append("%-44s dispatches=%6d live=%6d  FETCH_SIZE median %12.1f KB  WRITE_SIZE median %12.1f KB  -> %8.2f MB corrected"
                 % (k, nf, lf, f, w, (2 * f + w) * 1024.0 / 1e6))
# the instances tsgo_time_kernel launches (f64 slot planes, product mode), under the plain names bench.py's kernel table uses
for plain, ending in (("k_schur_lm", ", 0, 0>"), ("k_schur_pose", ", 0>"), ("k_cg_update", ">"), ("k_lin_lm", ">"), ("k_lin_pose", ">")):
    for k in list(kern):
        if k.startswith(plain + "<") and k.endswith(ending):
            kern[plain] = kern[k]
open(os.path.join(out_dir, tag + "_pmc_digest.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"workload": workload, "precision": 64, "source": tag + "_pmc_digest.txt", "kernels": kern},
          open(os.path.join(out_dir, tag + "_pmc_traffic.json"), "w"), indent=1)
print("\n".join(l for l in lines if "k_schur" in l or "k_lin" in l or "k_cg" in l))
