"""CPU oracle for the AMT-SAGA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a numpy restatement of the reference's inner transcription
loop (util_audio.audio_complete, RDCNN.res_net forward, the classifier heads'
classify(), the training.py feature recipe).  Every function cites the
reference file:line it follows (paths relative to the upstream repo).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product package
(``amt-saga_amd/amt_saga``) never imports it and fails loudly when the HIP
extension is missing.

Pinning status (see DESIGN.md "Oracle pinning"):
  * Hyperparams / check_shape / list_to_nd_array : pinned by golden vectors
    emitted by importing the reference's util_train_test.py
    (tests/golden/gen_golden_from_reference.py).
  * subtract / _resize / compress_bands / section / section_power /
    midi_tone_to_FFT / frame<->second maps : pinned by golden vectors emitted by
    the reference's own util_audio.audio_complete control flow (same script).
  * STFT -> magphase -> subtract -> iSTFT chain : pinned against the
    reference's recorded subtraction_demo FLAC triples (librosa output,
    PCM-24) via tests/golden/subtraction_demo_*.npz.
  * CQT (librosa.cqt) and the RDCNN forward (Keras): PARITY UNPINNED -- the
    reference holds no vector for either (no weights, no recorded outputs,
    librosa absent/unversioned).  The build defines the spec; see oracle/cqt.py
    and oracle/rdcnn.py headers.
"""
