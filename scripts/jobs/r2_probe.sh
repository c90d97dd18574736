python scripts/debug/fused_k8_probe.py
