for st in 0 2 4 6 8 10 12 16; do echo "stagger $st"; DP_W16_STAGGER=$st python3 tools/w16_sweep.py 32768 65536 2>&1 | grep "config 5"; done
