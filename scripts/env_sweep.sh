run() { echo "== $*"; env "$@" timeout -k 10 100 python scripts/spp_sweep.py 64 8 2>/dev/null | sed -e 's/{.*ms_extend/ms_extend/' -e 's/launches.*//'; }
run RT_X=0
run RT_TUNE=0,16,1,3,4,64
run RT_TUNE=0,32,1,3,4,64
run RT_TUNE=0,24,1,2,3,64
run RT_TUNE=0,24,1,4,5,64
run RT_TUNE=0,24,2,3,4,64
run RT_TUNE=64,24,1,3,4,64
run RT_TUNE=256,24,1,3,4,64
run RT_TUNE=0,24,1,3,4,32
run RT_MIX_PATHS=67108864
run RT_MIX_PATHS=268435456
run RT_ASYNC_BELOW=4194304
run RT_ASYNC_BELOW=67108864
