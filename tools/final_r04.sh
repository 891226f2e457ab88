#!/bin/bash
# Round 4 closing runs on one box: GPU test suite, rate probes, latency / spline probes, fuzz campaign, soak -- each into gpurun_out/r04/.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r04
python3 -m pytest tests -m gpu -q > gpurun_out/r04/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest.log; tail -3 gpurun_out/r04/gputest.log
python3 tools/bary_rate_probe.py > gpurun_out/r04/bary_rate_probe.txt 2>&1; echo "bary probe rc $?"
python3 tools/tt_rate_probe.py > gpurun_out/r04/tt_rate_probe.txt 2>&1; echo "tt probe rc $?"
python3 tools/latency_probe.py > gpurun_out/r04/latency.txt 2>&1; echo "latency rc $?"
python3 tools/spline_slider_probe.py > gpurun_out/r04/spline_slider_probe.txt 2>&1; echo "spline probe rc $?"
timeout -k 10 400 python3 tools/fuzz_campaign.py --seconds 300 > gpurun_out/r04/fuzz_campaign.txt 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r04/fuzz_campaign.txt
timeout -k 10 200 python3 tools/soak.py --seconds 90 > gpurun_out/r04/soak.txt 2>&1; echo "soak rc $?"; tail -2 gpurun_out/r04/soak.txt
timeout -k 10 100 python3 tools/leak_probe.py > gpurun_out/r04/leak_probe.txt 2>&1; echo "leak rc $?"
# a GPU fault anywhere above must fail the call whatever the individual exit codes were
if grep -l "Memory access fault\|GPU core dump" gpurun_out/r04/*.txt gpurun_out/r04/*.log > /dev/null 2>&1; then echo "GPU FAULT in: $(grep -l "Memory access fault" gpurun_out/r04/*.txt gpurun_out/r04/*.log)"; exit 1; fi
