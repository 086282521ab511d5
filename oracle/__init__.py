"""CPU oracle for the Asyrp DDIM hot path — TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain PyTorch-CPU fp32 restatement of the
reference algorithm (kwonminki/Asyrp_official, file:line cited per function).
It exists to *check* the HIP engine.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``asyrp_official_amd`` never does and fails loudly when its HIP library is missing.

Pinning status: the reference ships no tests or golden vectors.
The oracle is pinned instead against OUTPUTS OF THE REFERENCE ITSELF: the script
``tests/golden/make_golden.py`` imports the reference's own ``DDPM`` /
``denoising_step`` from ``/root/reference`` in the build container, runs them on
hash-generated weights/inputs and commits the results under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those fixtures.
"""
