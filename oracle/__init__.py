"""CPU oracle for the PMoE stage-2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pmoe_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.
It is the checker, never the thing measured or shipped.

Parity status: PINNED for everything the reference itself defines (``model/moe.py``,
``model/blocks/basics.py``, ``trainer/loss.py:121-132``) by golden vectors generated in the
build container from the imported reference (``oracle/make_golden.py`` ->
``tests/golden/*.pt``).  UNPINNED at one boundary: the ResNet-18 body comes from
``torchvision==0.9.1`` (``requirements.txt:97``), which is not vendored in the reference and not
installed here; ``oracle/resnet_topology.py`` restates its published topology and the reference's
own state_dict key/shape layout is the only pin for it.
"""
