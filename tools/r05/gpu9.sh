#!/bin/bash
# round 5, GPU call 9: how box-dependent is the tile table?  The tuner run again on another box, choices compared with the committed
# table (how many differ, and by how much the committed choice loses on THIS box's medians).
O=gpurun_out/r05_g9
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python tools/tune_tiles.py --out $O/tile_table_box2.json > $O/tune.log 2>&1
tail -2 $O/tune.log
python - <<'PY'
import json
a=json.load(open("neuralsvb_amd/tile_table.json")); b=json.load(open("gpurun_out/r05_g9/tile_table_box2.json"))
keys=[k for k in a["choices"] if k in b["choices"]]
diff=[k for k in keys if a["choices"][k]!=b["choices"][k]]
loss=[]
for k in diff:
    m=b["medians_us"][k]; ca=a["choices"][k]-1
    loss.append(m[ca]/min(m)-1.0)
loss.sort()
tot_best=sum(min(b["medians_us"][k]) for k in keys); tot_comm=sum(b["medians_us"][k][a["choices"][k]-1] for k in keys)
print(f"signatures in both: {len(keys)}; choices that differ: {len(diff)}; on this box the committed choice is slower than this box's own best by "
      f"median {100*loss[len(loss)//2]:.2f} % / worst {100*loss[-1]:.2f} % on those; summed over all signatures (one launch each): {100*(tot_comm/tot_best-1):.2f} %")
PY
