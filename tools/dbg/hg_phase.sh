#!/bin/bash
cd /root/repo
timeout 900 python tools/chunk_ab.py --profile 1 --crc --reps 1 "BSX_PHASES=1" 2>&1 | grep -v "^\[M::process\|^\[M::bsx_stream\|M::declined" | tail -40
