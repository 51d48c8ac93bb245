for nw in 4 8; do echo "masked sweep f32_nw=$nw"; FM_F32_NW=$nw FM_SELF_TRI=0 python scripts/gpu_f32_selfdist.py 100000 2>&1 | grep "self_tri 0"; done
