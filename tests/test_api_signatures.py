"""The drop-in boundary, pinned: every public function / class / method of the reference's section-8(a) modules exists here under the same name with
the same parameters (names, order, kinds, defaults).  ``tests/golden/api_signatures.json`` is ``inspect.signature`` of the reference's surface, dumped
by ``oracle/gen_golden.py api_signatures`` (data about signatures, no source); ``oracle.gen_golden.api_surface`` reads this package the same way.

What may differ, and nothing else:
* trailing extra parameters WITH defaults (``device=None``, kernel-side options) behind the reference's;
* the JAX pytree protocol (``tree_flatten`` / ``tree_unflatten``: SURVEY.md section 2 puts the JAX branches out of scope; jax is not in the image);
* the entries of ``KNOWN`` below, each with its reason.
"""
import json
import os

import pytest

from oracle.gen_golden import api_surface

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'api_signatures.json')
PYTREE = ('tree_flatten', 'tree_unflatten')
# (module, object[, member]) -> why it is not held to the reference's signature
KNOWN = {}


@pytest.fixture(scope='module')
def surfaces():
    with open(GOLDEN) as f:
        ref = json.load(f)
    return ref, api_surface('cosmoprimo_amd')


def compare(where, ref, here):
    """Differences between two parameter lists that a caller of the reference could notice."""
    if ref is None:
        return []
    if here is None:
        return ['{}: no signature here'.format(where)]
    ref = [p for p in ref if p[0] not in ('self', 'cls')]
    here = [p for p in here if p[0] not in ('self', 'cls')]
    out = []
    names_here = [p[0] for p in here]
    positional_ref = [p[0] for p in ref if p[1] == 'POSITIONAL_OR_KEYWORD']
    for name, kind, default in ref:
        if kind in ('VAR_POSITIONAL', 'VAR_KEYWORD'):
            if not any(p[1] == kind for p in here):
                out.append('{}: the reference takes {} ({}), not taken here'.format(where, '*args' if kind == 'VAR_POSITIONAL' else '**kwargs', name))
            continue
        if name not in names_here:
            out.append('{}: parameter {!r} of the reference is missing (reference {}, here {})'.format(where, name, [p[0] for p in ref], names_here))
            continue
        mine = here[names_here.index(name)]
        if kind == 'POSITIONAL_OR_KEYWORD' and (mine[1] != kind or names_here.index(name) != positional_ref.index(name)):
            out.append('{}: parameter {!r} is number {:d} in the reference, {} number {:d} here'.format(where, name, positional_ref.index(name), mine[1],
                                                                                                  names_here.index(name)))
        if kind == 'KEYWORD_ONLY' and mine[1] not in ('KEYWORD_ONLY', 'POSITIONAL_OR_KEYWORD'):
            out.append('{}: parameter {!r} cannot be given by keyword here'.format(where, name))
        if default != mine[2]:
            out.append('{}: default of {!r} is {} in the reference, {} here'.format(where, name, default, mine[2]))
    names_ref = {p[0] for p in ref}
    for name, kind, default in here:
        if name not in names_ref and kind in ('POSITIONAL_OR_KEYWORD', 'KEYWORD_ONLY') and default is None:
            out.append('{}: extra parameter {!r} without a default'.format(where, name))
    return out


def test_public_surface_matches_the_reference(surfaces):
    ref, here = surfaces
    problems = []
    for module, entries in ref.items():
        if module == '__init__':
            continue
        for name, entry in entries.items():
            if (module, name) in KNOWN:
                continue
            mine = here[module].get(name)
            if mine is None:
                problems.append('{}.{}: {} of the reference is missing'.format(module, name, entry['kind']))
                continue
            if mine['kind'] != entry['kind']:
                problems.append('{}.{}: {} in the reference, {} here'.format(module, name, entry['kind'], mine['kind']))
                continue
            if entry['kind'] == 'function':
                problems += compare('{}.{}'.format(module, name), entry['sig'], mine['sig'])
                continue
            problems += compare('{}.{}()'.format(module, name), entry['init'], mine['init'])
            for member, spec in entry['members'].items():
                if member in PYTREE or (module, name, member) in KNOWN:
                    continue
                got = mine['members'].get(member)
                where = '{}.{}.{}'.format(module, name, member)
                if got is None:
                    problems.append('{}: {} of the reference is missing'.format(where, spec['kind']))
                elif spec['kind'] in ('method', 'staticmethod', 'classmethod'):
                    if got['kind'] != spec['kind']:
                        problems.append('{}: {} in the reference, {} here'.format(where, spec['kind'], got['kind']))
                    else:
                        problems += compare(where, spec['sig'], got['sig'])
                elif spec['kind'] == 'property' and got['kind'] not in ('property', 'attribute'):
                    problems.append('{}: property in the reference, {} here'.format(where, got['kind']))
    assert not problems, '\n'.join(problems)


def test_star_import_gives_the_reference_names(surfaces):
    ref, here = surfaces
    assert here['__init__']['all'] == ref['__init__']['all']
    import cosmoprimo_amd
    for name in ref['__init__']['all']:
        assert hasattr(cosmoprimo_amd, name), name


def test_known_exceptions_still_exist(surfaces):
    """An entry of KNOWN that no longer names anything of the reference is stale."""
    ref, _ = surfaces
    for key in KNOWN:
        entry = ref[key[0]][key[1]]
        if len(key) == 3:
            assert key[2] in entry['members'], key


def test_section_getters_without_an_engine():
    """``Harmonic`` / ``Perturbations`` exist by name (no engine of this package has those sections); without an engine every getter asks for one,
    with the reference's exception (cosmology.py:662)."""
    import cosmoprimo_amd as cp
    from cosmoprimo_amd import cosmology
    cosmo = cp.Cosmology()
    for name in ('Background', 'Thermodynamics', 'Primordial', 'Perturbations', 'Transfer', 'Harmonic', 'Fourier'):
        with pytest.raises(cp.CosmologyInputError):
            getattr(cosmology, name)(cosmology=cosmo)
        with pytest.raises(cp.CosmologyInputError):
            getattr(cosmo, 'get_' + name.lower())()
