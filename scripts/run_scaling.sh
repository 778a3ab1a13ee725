#!/bin/bash
# The strong-scaling curve of the row-sharded EM loop on one node: bench.py at N = 1, 2, 4, 8 (whatever the node
# has), one JSON line each into <outdir>/scale_N.json, and a table of what explains the curve -- per N the sharded
# iteration, the rank-local compute alone, the exchange alone, per exchange candidate (config.exchange).
# usage: scripts/run_scaling.sh [outdir] [extra bench.py flags, e.g. --bootstraps 0 --cells 0]
#        SAME_DEVICE=1: all ranks on cuda:0 over the peer-to-peer exchange (a self test of the path, not scaling)
set -u
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/scaling}; shift || true
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python -c 'import torch; print(torch.cuda.device_count())')
same=""; [ "${SAME_DEVICE:-0}" = "1" ] && same="--same-device"
for n in 1 2 4 8; do
  if [ -z "$same" ] && [ "$n" -gt "$ngpu" ]; then break; fi
  if [ "$n" = 1 ]; then
    python bench.py --gpus 1 --steps 200 --warmup 20 "$@" > "$out/scale_$n.json" 2> "$out/scale_$n.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 200 --warmup 20 $same "$@" > "$out/scale_$n.json" 2> "$out/scale_$n.err"
  fi
  echo "N=$n rc=$?"
done
python - "$out" <<'PY'
import json, sys, os, glob
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "scale_*.json")), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    line = [l for l in open(f) if l.startswith("{")]
    if not line:
        continue
    j = json.loads(line[-1])
    ex = (j.get("config") or {}).get("exchange") or {}
    base = dict(n=j["n_gpus"], it_s=j["value"], ms=j["ms_per_step"], backend=(ex.get("backend") or "-").split(" (")[0],
                compute=ex.get("shard_compute_us"), rccl_ranks=ex.get("rccl_ranks_seen"))
    cands = {k: (v.get("iteration_us"), v.get("exchange_us")) for k, v in (ex.get("candidates") or {}).items() if v.get("ok")}
    rows.append((base, cands))
if rows:
    one = rows[0][0]["it_s"]
    print(f"{'N':>2} {'it/s':>9} {'x N=1':>6} {'eff':>5} {'ms/it':>7} {'shard compute us':>17} {'carried by':>14} {'rccl ranks':>10}  candidates: iteration us / exchange us")
    for b, c in rows:
        cs = "  ".join(f"{k}: {v[0]:.1f} / {v[1]:.1f}" for k, v in c.items())
        print(f"{b['n']:>2} {b['it_s']:9.0f} {b['it_s'] / one:6.2f} {b['it_s'] / one / b['n']:5.2f} {b['ms']:7.4f} "
              f"{(b['compute'] if b['compute'] is not None else float('nan')):17.1f} {b['backend']:>14} {str(b['rccl_ranks']):>10}  {cs}")
PY
