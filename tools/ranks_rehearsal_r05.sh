#!/bin/bash
# The multi-rank path rehearsed on ONE GPU with this round's code (PPO's prologue now crosses ranks between tg_ppo_returns and
# tg_ppo_norm): gloo ranks sharing the device with --check (N-rank vs one-rank agreement of trajectories, PPO's global moments and
# post-step weights inside the bench job) on C4's strong-scaling shard, and the driver's own torchrun line at world 1 with every
# collective issued through RCCL (TG_COLLECTIVES_AT_WORLD_1=1).  Eight ranks are not run here: the pool allows six GPU processes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
for n in 2 4; do
  timeout -k 10 400 python3 bench.py --gpus $n --backend gloo --check --config c4 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r05_bench_gloo_${n}ranks_check_c4.json 2> $OUT/ranks_gloo_$n.err || { echo "gloo $n failed"; tail -5 $OUT/ranks_gloo_$n.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/r05_bench_gloo_${n}ranks_check_c4.json')); print('gloo', $n, 'ranks:', round(d['value']/1e6,2), 'M env-steps/s, n_ranks_seen', d['n_ranks_seen'], 'check', d['rank_count_check'])"
done
TG_COLLECTIVES_AT_WORLD_1=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --other-configs off > $OUT/r05_bench_rccl_one_rank.json 2> $OUT/ranks_rccl.err || { echo "rccl failed"; tail -5 $OUT/ranks_rccl.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/r05_bench_rccl_one_rank.json')); print('rccl world 1:', round(d['value']/1e6,3), 'M env-steps/s; collectives', d['collectives']['backend'], d['collectives']['per_step'], 'per step,', d['collectives']['by_tag'])"
