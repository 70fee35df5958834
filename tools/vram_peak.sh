#!/bin/bash
# usage: tools/vram_peak.sh OUT -- command...   runs the command and writes the peak "VRAM Total Used Memory (B)" of GPU 0 (rocm-smi, sampled each second) to OUT
out=$1; shift; shift
( peak=0; while true; do u=$(rocm-smi --showmeminfo vram 2>/dev/null | awk '/Total Used Memory/ {print $NF; exit}'); [ -n "$u" ] && [ "$u" -gt "$peak" ] && peak=$u && echo $peak > "$out"; sleep 1; done ) &
sampler=$!
"$@"; rc=$?
kill $sampler 2>/dev/null
exit $rc
