set -e
cd /root/repo
mkdir -p gpurun_out/r5J
hipcc -std=c++17 -O1 -ffp-contract=off -I pred-occ-planner_amd/host tests/facade_replan_gpu_test.cpp -o /tmp/frt -L pred-occ-planner_amd -lsogm_hip -Wl,-rpath,/root/repo/pred-occ-planner_amd
fail=0
for i in $(seq 1 40); do
  if ! timeout 120 /tmp/frt > /tmp/frt_out.txt 2>&1; then fail=$((fail+1)); echo "run $i FAILED"; grep -n "agent\|delta\|REQUIRE" /tmp/frt_out.txt | head -12; fi
done
echo "failures: $fail of 40"
