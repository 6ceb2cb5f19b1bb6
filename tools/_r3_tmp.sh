timeout 900 python tools/big_sizes.py 2>&1 | grep -v Warn | tail -14
timeout 600 python tools/big_adjoint.py 2>&1 | grep -v Warn | tail -8
