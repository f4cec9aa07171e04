"""tools/collect_profiles.sh step: attach the HBM traffic / MFMA-busy figures of the dominant kernel (profiles/pmc_traffic.json, the
rocprofv3 --pmc passes tools/final_measure.sh ran behind the bench line on the same box) to the committed bench line -- bench.py reads
the same file at run time, but inside final_measure.sh the line is printed before the counter passes of that box exist."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1]
d = json.loads(open(path).read().strip().splitlines()[-1])
pmc = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
r = d.get("roofline") or {}
ent = pmc.get("variants", {}).get(r.get("kernel"))
if ent and pmc.get("batch") == d["config"]["per_gpu_batch"]:
    r["traffic"] = ent.get("bytes_per_launch")
    r["traffic_source"] = ("committed profile (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                           "the same box, run behind this line by tools/final_measure.sh; not measured inside the timed run)")
    r["traffic_note"] = pmc.get("note")
    r["mfma_busy"] = ent.get("mfma_busy")
    r["mfma_busy_over_sq_busy"] = ent.get("mfma_busy_over_sq_busy")
    r["algorithmic_bytes"] = ent.get("algorithmic_bytes_per_launch")
    # attribution: the counters' binary against the binary that printed the line (bench.py's `binary` field)
    b = d.get("binary") or {}
    same = pmc.get("source_sha256") is not None and pmc.get("source_sha256") == b.get("source_sha256")
    r["traffic_binary"] = {"profiled_lib_sha256": pmc.get("lib_sha256"), "profiled_source_sha256": pmc.get("source_sha256"),
                           "running_lib_sha256": b.get("lib_sha256"), "running_source_sha256": b.get("source_sha256"),
                           "same_sources": bool(same), "same_library_file": pmc.get("lib_sha256") == b.get("lib_sha256")}
    if same:
        r.pop("traffic_warning", None)
open(path, "w").write(json.dumps(d) + "\n")
print("traffic", r.get("traffic"), "mfma_busy", r.get("mfma_busy"))
