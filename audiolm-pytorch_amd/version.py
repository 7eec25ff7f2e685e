"""Checkpoint compatibility tag.  `*Transformer.load()` (reference audiolm_pytorch.py:625-640) compares the `version` field of a checkpoint
package with the running package's version and warns on older ones; `trainer.save()` writes it.  This drop-in reads and writes the
reference's checkpoints, so it reports the version of the reference release whose state_dict layout it mirrors
(tests/test_host_logic.py::test_state_dict_matches_reference), not a version of its own kernels."""
TRACKED_REFERENCE_RELEASE = (2, 4, 0)
__version__ = '.'.join(str(v) for v in TRACKED_REFERENCE_RELEASE)
