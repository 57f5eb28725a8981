"""Turns the per-kernel HBM traffic tables of scripts/pmc_iter.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
passes) into profiles/<round>_linearize_pmc.json, stamped with the hash of the kernel source the counters were taken on
(bench.py reports `roofline.traffic` from it only while the hash still matches).
usage: python scripts/make_pmc_json.py r02 cfg4=gpurun_out/r02a/iteration_hbm_traffic_cfg4.txt cfg5=..."""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_hash():
    src = open(os.path.join(ROOT, "opendlv-logic-cfsd18-sensation-slam_amd", "csrc", "gs_kernels.hip")).read()
    a = src.index("// ------------------------------------------------------------------ A5-A7")
    b = src.index("// landmark diagonal blocks from the per-(wave tile, landmark) partials")
    return hashlib.sha256(src[a:b].encode()).hexdigest()


if __name__ == "__main__":
    tag = sys.argv[1]
    out = dict(kernel="gs::k_linearize_ell", kernel_source_sha256=kernel_hash(),
               method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only (scripts/pmc_iter.sh), "
                      "mean per launch inside full Gauss-Newton iterations; FETCH_SIZE x2 as /opt/skills/guides/MI355X_MICROARCH.md "
                      "section HBM prescribes for coalesced streams on gfx950 (re-checked in round 1 with scripts/calib/fetch_calib: 1 GiB "
                      "streams at 4, 8 and 16 B per lane report 524298.5 KB read, 1048576.0 KB written); KB = 1024 bytes",
               workloads={})
    for arg in sys.argv[2:]:
        name, path = arg.split("=")
        for line in open(path):
            if "k_linearize_ell" in line:
                m = re.search(r"FETCH_SIZE\s+([\d.]+) KB.*WRITE_SIZE\s+([\d.]+) KB", line)
                f, w = float(m.group(1)), float(m.group(2))
                out["workloads"][name] = dict(FETCH_SIZE_KB_raw=f, WRITE_SIZE_KB=w, traffic_bytes_per_launch=int((2 * f + w) * 1024))
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_linearize_pmc.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))
