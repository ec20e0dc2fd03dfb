#!/bin/bash
mkdir -p gpurun_out/r4c63
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r4c63/pytest.txt 2>&1; tail -4 gpurun_out/r4c63/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
