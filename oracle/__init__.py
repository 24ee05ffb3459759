"""CPU oracle for the cosmoprimo hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a numpy/scipy restatement of the reference's
algorithms, written from the math and pinned against golden
vectors generated from the reference itself (``oracle/gen_golden.py`` ->
``tests/golden/*.npz``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the *checker*.  The product
package ``cosmoprimo_amd`` never imports it: the product path fails loudly when
the HIP library is missing.
"""
