#!/bin/bash
# stage x occupancy-cap matrix on the current box
python tools/ab.py --rounds 3 st8=SGW_OPTIONS=fast_wg_per_cu=0 st7=SGW_OPTIONS=fast_wg_per_cu=7 st6=SGW_OPTIONS=fast_wg_per_cu=6 st5=SGW_OPTIONS=fast_wg_per_cu=5 "ns7=SGW_OPTIONS=stage=0;fast_wg_per_cu=7"
