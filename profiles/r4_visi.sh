#!/bin/bash
# Round 4: visibility passes -- wall time of the three forms + rocprofv3 kernel stats of the per-camera and the batched form
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r4_visi; mkdir -p $O
python profiles/visi_profile.py --cams 200 --mode all > $O/visi_wall.json 2> $O/visi_wall.err; tail -1 $O/visi_wall.json
for mode in percam flags; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o visi -- python profiles/visi_profile.py --cams 40 --reps 1 --mode $mode > $O/prof_$mode.json 2> $O/prof_$mode.err
  f=$(find $O/prof_$mode -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/r4_visi_${mode}_kernel_stats.csv && head -14 $f
done
