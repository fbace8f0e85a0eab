set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/p3
python tools/launch_series.py 56 16384 100 1 2>&1 | grep -v amdgpu > gpurun_out/p3/series_events.txt
python tools/launch_series.py 56 16384 100 0 2>&1 | grep -v amdgpu > gpurun_out/p3/series_noevents.txt
python tools/launch_series.py 56 65536 100 0 2>&1 | grep -v amdgpu >> gpurun_out/p3/series_noevents.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p3/prof -o trace -- python3 tools/launch_series.py 56 16384 100 1 > gpurun_out/p3/series_prof.txt 2>&1
python tools/series_gaps.py gpurun_out/p3/prof k_energy_codelet 100 > gpurun_out/p3/gaps.txt
cat gpurun_out/p3/series_noevents.txt gpurun_out/p3/gaps.txt; tail -1 gpurun_out/p3/series_events.txt; tail -1 gpurun_out/p3/series_prof.txt
