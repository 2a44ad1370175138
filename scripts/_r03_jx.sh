R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python3 scripts/mb_jobs.py 2>&1 | tail -2
sed -i 's/  atb_jobs_k<H, false><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(t);   \/\/ f32-input MFMA (see launch_atb)/  atb_jobs_k<H, true><<<grid, kGroupThreads, atb_lds_bytes(H), st>>>(t);/' dualmessagepassing_amd/csrc/dmp_atb.hip
timeout 600 python3 scripts/mb_jobs.py 2>&1 | tail -2
sed -i 's/__global__ __launch_bounds__(kGroupThreads, 2) void atb_jobs_k/__global__ __launch_bounds__(kGroupThreads, 1) void atb_jobs_k/' dualmessagepassing_amd/csrc/dmp_atb.hip
timeout 600 python3 scripts/mb_jobs.py 2>&1 | tail -2
