mkdir -p gpurun_out/r3e
python -m pytest tests -m gpu -x -q -k "move or wide_samples or fuzz_wide or wide_rolling_kernel_full" > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3e/pytest.log; tail -3 gpurun_out/r3e/pytest.log
L=gpurun_out/r3e/sweep.log
# k_roll, one image per launch: tall tiles, deeper prefetch
python tools/ab.py --single --bpc 1 --bpp 4 --interp 2 --images 8 --rounds 3 --reps 8 HEAD th100,FIXCA_TH=100 th134,FIXCA_TH=134 th160,FIXCA_TH=160 th200,FIXCA_TH=200 th267,FIXCA_TH=267 th400,FIXCA_TH=400 \
   rd10@rd10,FIXCA_RING=22 rd10th100@rd10,FIXCA_RING=22,FIXCA_TH=100 rd10th160@rd10,FIXCA_RING=22,FIXCA_TH=160 rd10th200@rd10,FIXCA_RING=22,FIXCA_TH=200 rd10th267@rd10,FIXCA_RING=22,FIXCA_TH=267 \
   rd12th160@rd12,FIXCA_RING=24,FIXCA_TH=160 rd12th200@rd12,FIXCA_RING=24,FIXCA_TH=200 rd12th267@rd12,FIXCA_RING=24,FIXCA_TH=267 r2@r2head >> $L 2>&1
# k_move with the planner's defaults, single and batch, every format
for cfg in "2 4" "4 4" "-8 4" "2 3" "-4 3" "-8 3"; do set -- $cfg
  python tools/ab.py --single --bpc $1 --bpp $2 --interp 0 --images 6 --rounds 2 --reps 8 HEAD >> $L 2>&1
  python tools/ab.py --bpc $1 --bpp $2 --interp 0 --images 6 --rounds 2 --reps 8 HEAD,per=6 nu1,FIXCA_MOVE_NU=1,per=6 nu2,FIXCA_MOVE_NU=2,per=6 nu4,FIXCA_MOVE_NU=4,per=6 nu4th32,FIXCA_MOVE_NU=4,FIXCA_TH=32,per=6 nu2th64,FIXCA_MOVE_NU=2,FIXCA_TH=64,per=6 >> $L 2>&1
done
grep -v amdgpu.ids $L | cut -c1-230
