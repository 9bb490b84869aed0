"""The reference's four matplotlib training callbacks (gpsa/plotting/callbacks.py:17, 179, 321, 392) are OUT OF SCOPE
(plotting): the names resolve so that a caller's import block runs unchanged, a call says
where to get the real thing.  No plotting code lives here."""


def _out_of_scope(name):
    def callback(*args, **kwargs):
        raise NotImplementedError(
            f"gpsa.plotting.{name} is out of scope of the MI355X hot path (SURVEY.md section 2 row 9): "
            f"import it from the reference's gpsa.plotting.callbacks - it only reads model attributes and "
            f"forward()'s outputs, which this package provides unchanged")

    callback.__name__ = callback.__qualname__ = name
    return callback


callback_oned = _out_of_scope("callback_oned")
callback_twod = _out_of_scope("callback_twod")
callback_twod_aligned_only = _out_of_scope("callback_twod_aligned_only")
callback_twod_multimodal = _out_of_scope("callback_twod_multimodal")
