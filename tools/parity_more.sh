#!/bin/bash
# more samples of the two Env03-v2 campaign workloads with other seeds (GPU box; ~10 min per seed on 64 host threads):
#   bash tools/parity_more.sh SEED [outdir]     -> outdir/r03_parity_config3_large_seedS.json, _config3_policy_seedS.json
S=$1; O=${2:-gpurun_out/parity_more}; mkdir -p $O
python tools/parity_locate.py --student gpu --teacher oracle --seed $S --out $O/r03_parity_config3_large_seed$S.json --env Env03-v2 --envs 4096 --steps 500 --actions random --auto-reset 1 > $O/large_seed$S.log 2>&1
grep -E '"over"|"worst"' $O/r03_parity_config3_large_seed$S.json
python tools/parity_locate.py --student gpu --teacher oracle --seed $S --out $O/r03_parity_config3_policy_seed$S.json --env Env03-v2 --envs 2048 --steps 500 --actions policy --auto-reset 1 > $O/policy_seed$S.log 2>&1
grep -E '"over"|"worst"' $O/r03_parity_config3_policy_seed$S.json
