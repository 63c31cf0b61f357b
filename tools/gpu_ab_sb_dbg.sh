for D in 0 2 34 130 3 64; do echo "dbg=$D"; AOMHIP_SB_DBG=$D timeout 200 python tools/gpu_ab_sadsb.py 4k 8 64 480,32 2>&1 | tail -1; done
