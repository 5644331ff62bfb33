# how profiles/r04_psnr_study.json was made: the aux sections + six calls of 22 seeds each behind the first call of 16 (python tools/psnr_study.py ... 16),
# one gpurun call at a time, merged by tools/psnr_merge.py (run from /root/repo)
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 900 -- 'GFN_STUDY_AUX_ONLY=1 python tools/psnr_study.py gpurun_out/r04_psnr_study_aux.json 4 > gpurun_out/r04_psnr_study_aux.log 2>&1; tail -2 gpurun_out/r04_psnr_study_aux.log' > gpurun_out/_study_aux.out 2>&1
for K in 16 38 60 82 104 126; do
  /usr/local/graft/bin/gpurun --timeout 1190 -- "GFN_STUDY_SEED0=$K python tools/psnr_study.py gpurun_out/r04_psnr_study_p$K.json 22 > gpurun_out/r04_psnr_study_p$K.log 2>&1; tail -1 gpurun_out/r04_psnr_study_p$K.log" > gpurun_out/_study_p$K.out 2>&1
  tail -2 gpurun_out/_study_p$K.out
done
