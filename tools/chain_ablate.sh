for d in 0 4 5 2 3 1; do echo "== DG_CHAIN_DBG=$d"; DG_CHAIN_DBG=$d timeout -k 10 120 python tools/chain_check.py --mode 0 2>&1 | grep "M=16384" | tail -1 || exit 1; done
