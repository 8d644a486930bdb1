"""ORACLE package: CPU restatements used only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under omfs_4d_video_gen_amd/ imports it."""
