#!/bin/bash
# The GPU calls of a round, one stage per gpurun call (a lost box then costs one stage, not the lot).  A gpurun box: 256 cores, 3 TB of memory under a cgroup limit of 300 GiB,
# 79 GB of scratch space, one MI355X (profiles/round5_box.txt).  Round 5 ran every stage (profiles/round5_*); `unrun` is kept for tests that are written without a device at hand.
#   tools/round_start.sh box        what the box is: memory and its cgroup limit, scratch space, devices  (seconds; run it FIRST and keep it in front of every other stage:
#                                   three boxes were lost in round 4 and nothing is known about their limits)
#   tools/round_start.sh suite      python -m pytest tests -x -q -m gpu   (as the driver runs it)
#   tools/round_start.sh unrun      the four GPU tests that have not run yet (MM_TEST_NOT_YET_RUN=1), one by one, no -x
#   tools/round_start.sh scale      several device contexts at full size on the one GPU (MM_TEST_CONTEXTS_AT_SCALE=1; in two of the three calls that lost their box in round 4)
#   tools/round_start.sh profiles   tools/round_profiles.sh gpurun_out/round <tag>   (bench line, rocprofv3 stats, PMC passes, lane trace)
# Everything lands under gpurun_out/start/.
STAGE=${1:-box}; TAG=${2:-round5}; OUT=gpurun_out/start; mkdir -p "$OUT"; export TMPDIR=/tmp

box() {
	{
		echo "== memory"; free -g | head -2
		echo "== cgroup memory limit"; cat /sys/fs/cgroup/memory.max 2> /dev/null || cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2> /dev/null
		echo "== cgroup memory in use"; cat /sys/fs/cgroup/memory.current 2> /dev/null || cat /sys/fs/cgroup/memory/memory.usage_in_bytes 2> /dev/null
		echo "== scratch space"; df -h /tmp /root/repo / /dev/shm 2> /dev/null
		echo "== cores"; nproc; echo "== pids limit"; cat /sys/fs/cgroup/pids.max 2> /dev/null
		echo "== devices"; rocm-smi --showmeminfo vram 2> /dev/null | grep -i "total\|used" | head -16
	} > "$OUT/${TAG}_box.txt" 2>&1
	cat "$OUT/${TAG}_box.txt"
}

# a memory / scratch watch beside a stage: one line every 5 s, so that the last lines of a stage that loses its box are at least in the part of the log that was flushed
watch_box() { while true; do echo "$(date +%T) mem_used_gb=$(free -g | awk 'NR==2{print $3}') cgroup=$(cat /sys/fs/cgroup/memory.current 2> /dev/null) tmp_used=$(df --output=used -BG /tmp | tail -1)"; sleep 5; done; }

case "$STAGE" in
box) box ;;
suite)
	box > /dev/null; watch_box > "$OUT/${TAG}_suite_watch.txt" & W=$!
	python -m pytest tests -x -q -m gpu --durations=15 > "$OUT/${TAG}_suite.log" 2>&1; echo "rc=$?" >> "$OUT/${TAG}_suite.log"
	kill $W; tail -25 "$OUT/${TAG}_suite.log" ;;
unrun)
	box > /dev/null
	for t in test_bench_line_with_two_devices_in_one_process test_deferred_rescue_rounds_give_the_same_bytes test_extension_trials_that_start_at_the_end_of_a_section test_every_read_runs_with_the_value_the_reference_would_carry; do
		MM_TEST_NOT_YET_RUN=1 timeout 900 python -m pytest "tests/test_zz_bench_devices_gpu.py::$t" -q > "$OUT/${TAG}_unrun_$t.log" 2>&1; echo "$t rc=$?" | tee -a "$OUT/${TAG}_unrun.txt"
	done ;;
scale)
	box > /dev/null; watch_box > "$OUT/${TAG}_scale_watch.txt" & W=$!
	MM_TEST_CONTEXTS_AT_SCALE=1 timeout 2400 python -m pytest tests/test_headline_gpu.py -x -q --durations=10 > "$OUT/${TAG}_scale.log" 2>&1; echo "rc=$?" >> "$OUT/${TAG}_scale.log"
	kill $W; tail -15 "$OUT/${TAG}_scale.log" ;;
profiles) bash tools/round_profiles.sh gpurun_out/round "$TAG" ;;
*) echo "unknown stage $STAGE"; exit 2 ;;
esac
