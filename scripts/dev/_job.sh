python3 scripts/dev/time_kr_batch.py 5 100 gpurun_out/kr_la.npy 2>&1 | grep -v amdgpu
WDG_KR_KERNEL=blocked python3 scripts/dev/time_kr_batch.py 5 100 gpurun_out/kr_old.npy 2>&1 | grep -v amdgpu
python3 -c "
import numpy as np
a,b=np.load('gpurun_out/kr_la.npy'),np.load('gpurun_out/kr_old.npy')
print('hit counts equal:', np.array_equal(a,b))
"
export WDG_LIB_PATH=$PWD/when-do-gnns-help_amd/lib/variants/libwdg_hip_kr_prof.so
python3 scripts/dev/time_kr_batch.py 1 20 2>&1 | grep "k2 cycles" | tail -2
