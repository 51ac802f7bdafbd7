"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the detection hot path (conv graph -> head decode -> candidate filter -> NMS).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the shipped package ``yolo_amd`` never does.
"""
