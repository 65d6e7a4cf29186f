"""BMS group algebra (scri/bms_transformations.py): placeholder, filled in by a later milestone."""


class LorentzTransformation:  # pragma: no cover
    def __init__(self, **kwargs):
        raise NotImplementedError("LorentzTransformation is not implemented yet")


class BMSTransformation:  # pragma: no cover
    def __init__(self, **kwargs):
        raise NotImplementedError("BMSTransformation is not implemented yet")
