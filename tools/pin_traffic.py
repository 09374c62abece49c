"""Write profiles/traffic.json from a PMC summary (tools/pmc_summary.py output): the k_syrk_lower HBM bytes of one step,
pinned to the SHA-256 of the kernel source they were measured on and, for whole-evaluation figures, to the library's build id
(bench.py reports roofline.traffic only while the pin matches).
usage: python tools/pin_traffic.py profiles/<round>_batched128_pmc_traffic.json [N M chains [grad 0|1 [workload chain|subjects]]]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonstationary_multivariate_gaussian_process_amd import build  # noqa: E402
CHOL = os.path.join(ROOT, "nonstationary_multivariate_gaussian_process_amd", "csrc", "nmgp_chol.hip")


def main():
    src = sys.argv[1]
    N, M, chains = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2048, 3, 128)
    grad = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
    workload = sys.argv[6] if len(sys.argv) > 6 else "chain"
    summ = json.load(open(src))["summary"]
    out = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(out))
    rel = os.path.relpath(os.path.abspath(src), ROOT)
    entry = {"N": N, "M": M, "chains": chains, "grad": grad, "workload": workload,
             # the value line's roofline names k_syrk_lower: its launches; the end-to-end rooflines (value+gradient step, the
             # multi-subject factorisation stage) get the bytes of the WHOLE evaluation
             "bytes_per_step": (summ["k_syrk_lower"]["hbm_bytes_gfx950_corrected"] if (not grad and workload == "chain")
                                else summ["hbm_bytes_gfx950_corrected"]),
             "scope": "k_syrk_lower launches" if (not grad and workload == "chain") else "every kernel of the evaluation",
             "source": ("%s (per K class: %s)" % (rel, rel.replace("pmc_traffic", "pmc_syrk_classes"))
                        if os.path.exists(os.path.join(ROOT, rel.replace("pmc_traffic", "pmc_syrk_classes"))) else rel),
             "chol_sha256": hashlib.sha256(open(CHOL, "rb").read()).hexdigest(),
             # whole-evaluation figures belong to the whole build (bench.measured_traffic checks this one for them)
             "tree_id": build.tree_id()}
    doc["entries"] = [e for e in doc["entries"]
                      if (e["N"], e["M"], e["chains"], bool(e.get("grad", False)), e.get("workload", "chain")) != (N, M, chains, grad, workload)]
    doc["entries"].append(entry)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
