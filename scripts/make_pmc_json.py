"""Turns the per-kernel HBM traffic tables of scripts/pmc_iter.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
passes) into profiles/<round>_linearize_pmc.json, stamped with the hash of the kernel source the counters were taken on
(bench.py reports `roofline.traffic` from it only while the hash still matches).
usage: python scripts/make_pmc_json.py r02 cfg4=gpurun_out/r02a/iteration_hbm_traffic_cfg4.txt cfg5=..."""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def solver_hash():
    """hash of the factorisation / backward-solve kernels (variant 3) the solver traffic was measured on"""
    src = open(os.path.join(ROOT, "opendlv-logic-cfsd18-sensation-slam_amd", "csrc", "gs_kernels.hip")).read()
    a = src.index("// ---- variant 3: latency-shaped wave-per-front kernels")
    b = src.index("// ---- structure phase on the device: the ELL streams of the observation edges")
    return hashlib.sha256(src[a:b].encode()).hexdigest()


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
    # the solver kernels of the same passes: factor (leaf + tree launch) and backward solve (tree launch + leaf levels)
    sol = dict(kernels="gs::k_factor3 (leaf + tree launches), gs::k_backsolve3 (tree launch + leaf levels)", solver_source_sha256=solver_hash(),
               method=out["method"], workloads={})
    for arg in sys.argv[2:]:
        name, path = arg.split("=")
        tot, per = 0.0, {}
        for line in open(path):
            if line.startswith(("k_factor3", "k_backsolve3")):
                m = re.search(r"launches\s+(\d+)\s+FETCH_SIZE\s+([\d.]+) KB.*WRITE_SIZE\s+([\d.]+) KB", line)
                n, f, w = int(m.group(1)), float(m.group(2)), float(m.group(3))
                per[line.split("(")[0].strip()] = dict(launches_in_sample=n, FETCH_SIZE_KB_raw=f, WRITE_SIZE_KB=w)
        its = min(v["launches_in_sample"] for v in per.values()) if per else 0     # one launch per iteration for the tree / leaf-factor kernels
        for k, v in per.items():
            tot += (2 * v["FETCH_SIZE_KB_raw"] + v["WRITE_SIZE_KB"]) * 1024 * (v["launches_in_sample"] / its)
        if per:
            sol["workloads"][name] = dict(per_kernel=per, iterations_in_sample=its, traffic_bytes_per_iteration=int(tot))
    json.dump(sol, open(os.path.join(ROOT, "profiles", "%s_solver_pmc.json" % tag), "w"), indent=1)
    print(json.dumps(sol, indent=1))
