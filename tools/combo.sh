#!/bin/bash
# stage x occupancy-cap matrix on the current box
python tools/ab.py --rounds 3 st8=SGW_FAST_WG_PER_CU=0 st7=SGW_FAST_WG_PER_CU=7 st6=SGW_FAST_WG_PER_CU=6 st5=SGW_FAST_WG_PER_CU=5 ns8=SGW_NO_STAGE=1,SGW_FAST_WG_PER_CU=0 ns7=SGW_NO_STAGE=1,SGW_FAST_WG_PER_CU=7 ns6=SGW_NO_STAGE=1,SGW_FAST_WG_PER_CU=6
