timeout 1200 python tools/fuzz_build.py 300 2000 2>&1 | tail -3
LATTICE_ROW_ORDER=canonical timeout 1200 python tools/fuzz_build.py 150 500 2>&1 | tail -3
timeout 600 python tools/big_adjoint.py 2>&1 | tail -3
timeout 600 python tools/leak_check.py 2>&1 | tail -3
