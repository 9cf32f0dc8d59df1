"""Error types mirroring pyrado.utils.exceptions (TypeErr / ValueErr / ShapeErr / KeyErr), P/utils/exceptions.py."""


class BaseErr(Exception):
    pass


class TypeErr(BaseErr, TypeError):
    def __init__(self, *, given=None, expected_type=None, msg=None):
        if msg is None:
            msg = f"Expected the type {expected_type} but received {type(given)}!"
        super().__init__(msg)


class ValueErr(BaseErr, ValueError):
    def __init__(self, *, given=None, given_name=None, eq_constraint=None, l_constraint=None, le_constraint=None,
                 g_constraint=None, ge_constraint=None, msg=None):
        if msg is None:
            cons = dict(eq=eq_constraint, l=l_constraint, le=le_constraint, g=g_constraint, ge=ge_constraint)
            cons = ", ".join(f"{k} {v}" for k, v in cons.items() if v is not None)
            msg = f"The given value {given} violates the constraint: {cons}!"
        super().__init__(msg)


class ShapeErr(BaseErr):
    def __init__(self, *, given=None, expected_match=None, msg=None):
        if msg is None:
            gs = getattr(given, "shape", None)
            es = getattr(expected_match, "shape", expected_match)
            msg = f"The given shape {gs} does not match the expected shape {es}!"
        super().__init__(msg)


class KeyErr(BaseErr, KeyError):
    def __init__(self, *, keys=None, container=None, msg=None):
        if msg is None:
            msg = f"The key(s) {keys} was/were not found in {type(container)}!"
        super().__init__(msg)
