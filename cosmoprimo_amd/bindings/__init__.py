"""Adapters that let the samplers the reference plugs into (Cobaya, CosmoSIS) drive this package -- the batch callers of SURVEY.md 8(f) f4
(reference bindings/cobaya/cosmoprimo.py, bindings/cosmosis/cosmoprimo_interface.py).  Neither framework is imported at package import:
``cosmoprimo_amd.bindings.cobaya`` needs Cobaya, ``cosmoprimo_amd.bindings.cosmosis`` runs with or without CosmoSIS installed (its DataBlock is
only ever used through the object the framework hands over)."""
