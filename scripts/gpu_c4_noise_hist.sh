#!/bin/bash
# N cold processes of scripts/c4_gradient_noise.py on this box; the distribution of dL_dscales against round 4's bar.
# usage: gpu_c4_noise_hist.sh tag [N]
out=gpurun_out/${1:-c4hist}; N=${2:-30}; mkdir -p $out; export TMPDIR=/tmp
rm -f /tmp/c4_oracles.npz $out/runs.jsonl
for i in $(seq 1 $N); do timeout 120 python scripts/c4_gradient_noise.py atomic 2>/dev/null | grep '^{' >> $out/runs.jsonl; done
for i in 1 2 3; do timeout 120 python scripts/c4_gradient_noise.py deterministic 2>/dev/null | grep '^{' >> $out/runs.jsonl; done
python - "$out/runs.jsonl" <<'PY' | tee $out/summary.txt
import json, sys, socket
rows = [json.loads(l) for l in open(sys.argv[1])]
at = [r for r in rows if r["mode"] == "atomic"]; de = [r for r in rows if r["mode"] == "deterministic"]
print(f"box {socket.gethostname()}: {len(at)} cold processes, default mode (float atomics); round 4's comparison: max|hip - oracle_fp32| / max|oracle|, bar 2e-4")
for k in ("dL_dscales", "dL_drotations", "dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dcolors"):
    v = sorted(r[k]["vs_oracle"] for r in at); t = sorted(r[k]["vs_truth"] for r in at)
    over = sum(x > 2e-4 for x in v)
    print(f"  {k:14s} vs oracle: min {v[0]:.3e} median {v[len(v)//2]:.3e} max {v[-1]:.3e}   over 2e-4: {over}/{len(v)}    vs truth: {t[0]:.3e} .. {t[-1]:.3e}")
v = [r["dL_dscales"]["vs_oracle"] for r in at]
edges = [0.5e-4 * i for i in range(1, 13)]
print("  dL_dscales vs oracle, histogram (bin upper edge x 1e-4: count): " + ", ".join(f"{e * 1e4:.1f}: {sum((e - 0.5e-4) < x <= e for x in v)}" for e in edges) + f", above: {sum(x > edges[-1] for x in v)}")
print("  worst Gaussian of dL_dscales per run:", sorted(set(r["dL_dscales"]["worst_gaussian_vs_oracle"] for r in at)))
print("  deterministic mode, 3 cold processes, dL_dscales vs oracle:", [f'{r["dL_dscales"]["vs_oracle"]:.6e}' for r in de])
PY
