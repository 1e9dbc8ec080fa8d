mkdir -p gpurun_out/r3g
L=gpurun_out/r3g/sweep.log
python tools/ab.py --single --bpc 1 --bpp 4 --interp 2 --images 8 --rounds 3 --reps 8 HEAD th20,FIXCA_TH=20 th24,FIXCA_TH=24 th29,FIXCA_TH=29 th33,FIXCA_TH=33 th37,FIXCA_TH=37 th41,FIXCA_TH=41 th45,FIXCA_TH=45 th49,FIXCA_TH=49 th53,FIXCA_TH=53 th61,FIXCA_TH=61 th67,FIXCA_TH=67 th80,FIXCA_TH=80 HEAD2 >> $L 2>&1
python tools/ab.py --single --bpc 2 --bpp 4 --interp 0 --images 6 --rounds 2 --reps 8 --grid "FIXCA_MOVE_NU=1,2;FIXCA_MOVE_DEPTH=6,10,14;FIXCA_TH=64,100,160,270" >> $L 2>&1
python tools/ab.py --single --bpc 2 --bpp 3 --interp 0 --images 6 --rounds 2 --reps 8 --grid "FIXCA_MOVE_NU=1,2;FIXCA_MOVE_DEPTH=6,10,14;FIXCA_TH=64,100,160,270" >> $L 2>&1
python tools/ab.py --single --bpc -4 --bpp 3 --interp 0 --images 6 --rounds 2 --reps 8 --grid "FIXCA_MOVE_NU=1,2;FIXCA_MOVE_DEPTH=6,10,14;FIXCA_TH=64,100,160,270,400" >> $L 2>&1
python tools/ab.py --single --bpc -8 --bpp 3 --interp 0 --images 6 --rounds 2 --reps 8 --grid "FIXCA_MOVE_NU=2,4;FIXCA_MOVE_DEPTH=4,7,10;FIXCA_TH=64,100,160,270,400" >> $L 2>&1
python tools/ab.py --single --bpc -4 --bpp 4 --interp 0 --images 6 --rounds 2 --reps 8 --grid "FIXCA_MOVE_NU=1;FIXCA_MOVE_DEPTH=6,10,14;FIXCA_TH=400,800,1000,1334" >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-230
