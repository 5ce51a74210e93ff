"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MAG-BERT / MAG-XLNet hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline -- never as a fallback for the
HIP path (``bert_multimodal_transformer_amd`` raises if its HIP library is missing).

Contents
--------
weights.py       deterministic counter-hash weight/batch generator (numpy only)
mag_bert_ref.py  pure-torch fp32 restatement of MAG (modeling.py:6-51), the MAG-BERT
                 wrappers (bert.py:76-324) and the transformers==3.0.2 layers they call
optim_ref.py     transformers==3.0.2 AdamW + get_linear_schedule_with_warmup restated
make_golden.py   (survey container only) imports /root/reference through a shim,
                 checks the restatement against it and writes tests/golden/*.npz

Parity pin: the restatement is pinned by golden vectors produced by the reference's own
Python (modeling.py / bert.py executed verbatim) run in the build container;
the third-party layers under it came from transformers 5.15 configured to the 3.0.2
arithmetic (eager attention, -10000.0 mask) -- see make_golden.py.  AdamW has no
installed 3.0.2 copy: it is pinned by a hand-derived known-answer test ("parity
unpinned by a library", stated in DESIGN.md).
"""
