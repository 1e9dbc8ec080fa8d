mkdir -p gpurun_out/r3d
python -m pytest tests -m gpu -x -q -k "move or wide_samples or fuzz_wide" > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log; tail -3 gpurun_out/r3d/pytest.log
L=gpurun_out/r3d/sweep.log
# single image per launch: RGBA u16 / float / double
python tools/ab.py --single --bpc 2 --bpp 4 --interp 0 --images 8 --rounds 2 --reps 8 --grid "lib=HEAD,rb1,rbx,d10,d14;FIXCA_MOVE_NU=1,4;FIXCA_TH=100,160,270,400,800" >> $L 2>&1
python tools/ab.py --single --bpc -4 --bpp 4 --interp 0 --images 8 --rounds 2 --reps 8 --grid "lib=HEAD,rb1,rbx,d10,d14;FIXCA_MOVE_NU=1,2,4;FIXCA_TH=32,112,200,400,800" >> $L 2>&1
python tools/ab.py --single --bpc -8 --bpp 4 --interp 0 --images 6 --rounds 2 --reps 8 --grid "lib=HEAD,rb1,rbx,d10,d14;FIXCA_MOVE_NU=2,4;FIXCA_TH=32,130,270,400,800" >> $L 2>&1
# batch entry, 8 images per launch
python tools/ab.py --bpc 2 --bpp 4 --interp 0 --images 8 --rounds 2 --reps 8 --grid "lib=HEAD,d10;FIXCA_MOVE_NU=1,4;FIXCA_TH=32,64,100,160,400;per=8" >> $L 2>&1
python tools/ab.py --bpc -4 --bpp 4 --interp 0 --images 8 --rounds 2 --reps 8 --grid "lib=HEAD,d10;FIXCA_MOVE_NU=1,2,4;FIXCA_TH=32,64,112,200,400;per=8" >> $L 2>&1
python tools/ab.py --bpc -8 --bpp 4 --interp 0 --images 6 --rounds 2 --reps 8 --grid "lib=HEAD,d10;FIXCA_MOVE_NU=2,4;FIXCA_TH=32,64,130,270,400;per=6" >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-210
