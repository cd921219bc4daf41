"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference (restate.py), the shim that runs the real reference here
(ref_shim.py) and the generator of tests/golden/ (gen_golden.py).  Nothing under real-routing-nco_amd/ may import this
package; only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py do."""
