cd $GRAFT_REPO_ROOT
GFN_STUDY_AUX_ONLY=1 timeout -k 10 600 python tools/psnr_study.py gpurun_out/r04_psnr_study_aux.json 4 > gpurun_out/r04_psnr_study_aux.log 2>&1 || exit 1
grep "gradient cosine" gpurun_out/r04_psnr_study_aux.log | cut -c1-1500
A="--steps 10 --no-cpu-baseline --global-steps 0 --psnr-steps 0,0 --trained-seeds= --render-frames 0"
for P in 0 1 0 1; do
  GFN_SIDE_PRIORITY=$P timeout -k 10 200 python bench.py $A 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side priority $P: headline', d['ms_per_step'], 'plugin', d['config'].get('plugin_ms_per_step'), d['config'].get('plugin_global_ms_per_step'))"
done
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
