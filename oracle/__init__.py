"""CPU oracle for the FutureDet LiDAR hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker.  futuredet_amd/ never imports it.
"""
