# The C++ replan transcription test after the LDS of every CU has been filled with a pattern (tools/micro/lds_poison.hip):
# a read of uninitialised LDS in either path shows up as a mismatch.  bash tools/loop_facade_lds.sh
cd /root/repo
hipcc --offload-arch=gfx950 -O2 tools/micro/lds_poison.hip -o /tmp/lds_poison || exit 1
hipcc -std=c++17 -O1 -ffp-contract=off -I pred-occ-planner_amd/host tests/facade_replan_gpu_test.cpp -o /tmp/frt -L pred-occ-planner_amd -lsogm_hip -Wl,-rpath,/root/repo/pred-occ-planner_amd || exit 1
fail=0; n=0
for rep in 1 2 3; do
for pat in 0xffffffff 0x7ff80000 0x7fc00000 0xdeadbeef 0x00000001 0x3f800000 0x80000000 0x7fffffff; do
  /tmp/lds_poison $pat > /dev/null || echo "poison failed"
  n=$((n+1))
  if ! timeout 120 /tmp/frt > /tmp/frt_out.txt 2>&1; then fail=$((fail+1)); echo "pattern $pat FAILED"; grep -n "agent\|delta\|REQUIRE" /tmp/frt_out.txt | head -8; fi
done
done
echo "failures: $fail of $n"
