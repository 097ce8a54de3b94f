#!/bin/bash
# what kind of box is this?  (k_seed runs 3-4x slower on some boxes of the pool)
rocm-smi --showmemorypartition --showcomputepartition --showclocks --showperflevel 2>/dev/null | grep -v "^$" | head -40
rocm-smi --showmeminfo vram 2>/dev/null | grep -v "^$" | head -6
cat /sys/class/drm/card*/device/current_memory_partition 2>/dev/null | head -2
cat /sys/class/drm/card*/device/current_compute_partition 2>/dev/null | head -2
cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E "simd_count|mem_banks|cu_per_simd|max_engine_clk|num_xcc|array_count" | sort | uniq -c | head
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
python bench.py --genome-mbp 128 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K_SEED_MS', d['kernel_ms_per_step']['seed'], 'TIER1', d['kernel_ms_per_step']['regions_tier1'], d['device'])"
