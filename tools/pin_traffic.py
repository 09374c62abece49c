"""Write profiles/traffic.json from a PMC summary (tools/pmc_summary.py output): the k_syrk_lower HBM bytes of one step,
pinned to the SHA-256 of the kernel source they were measured on (bench.py reports roofline.traffic only while it matches).
usage: python tools/pin_traffic.py profiles/<round>_batched128_pmc_traffic.json [N M chains]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHOL = os.path.join(ROOT, "nonstationary_multivariate_gaussian_process_amd", "csrc", "nmgp_chol.hip")


def main():
    src = sys.argv[1]
    N, M, chains = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2048, 3, 128)
    summ = json.load(open(src))["summary"]
    out = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(out))
    rel = os.path.relpath(os.path.abspath(src), ROOT)
    entry = {"N": N, "M": M, "chains": chains, "grad": False,
             "bytes_per_step": summ["k_syrk_lower"]["hbm_bytes_gfx950_corrected"],
             "source": "%s (per K class: %s)" % (rel, rel.replace("pmc_traffic", "pmc_syrk_classes")),
             "chol_sha256": hashlib.sha256(open(CHOL, "rb").read()).hexdigest()}
    doc["entries"] = [e for e in doc["entries"] if (e["N"], e["M"], e["chains"], e.get("grad", False)) != (N, M, chains, False)]
    doc["entries"].append(entry)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(entry))


if __name__ == "__main__":
    main()
