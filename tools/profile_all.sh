set -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/profile.sh r2_snake --steps 100 --warmup 10 > gpurun_out/pa_snake.log 2>&1; echo snake $?
bash tools/profile.sh r2_crypto --workload crypto_1m --steps 12 --warmup 3 > gpurun_out/pa_crypto.log 2>&1; echo crypto $?
bash tools/profile.sh r2_traffic --workload traffic_262k --steps 40 --warmup 5 > gpurun_out/pa_traffic.log 2>&1; echo traffic $?
for w in parking climate fleet manufacturing hospital; do bash tools/profile.sh r2_$w --workload ${w}_131k --steps 40 --warmup 5 > gpurun_out/pa_$w.log 2>&1; echo $w $?; done
