#!/bin/bash
timeout 100 python tools/dbg/fused_diag.py 3000000 2>&1 | tail -60 | cut -c1-400
