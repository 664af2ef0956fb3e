timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo rc=$? >> gpurun_out/t.log; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" gpurun_out/t.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/collect_evidence.sh > gpurun_out/evidence.log 2>&1; tail -1 gpurun_out/evidence.log
