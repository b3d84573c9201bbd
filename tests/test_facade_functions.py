"""The problem façade's function container and the adapter's equality / inequality split (reference python/prb.py:166-204,
:379-402; python/ddp.py:38-48, :108-111): names, order and node ranges follow the reference's literals; the analytic model's
constants are derived from what the container holds.  CPU only (the adapter object is built without its engine)."""
import numpy as np
import pytest

from srbd_horizon_amd.ddp import DDPSolver, MODEL_TERMS
from srbd_horizon_amd.prb import LIPProblem, SRBD13Problem, SRBDProblem
from srbd_horizon_amd.problem import Term

NS = 20


def _adapter(prb):
    s = DDPSolver.__new__(DDPSolver)            # the adapter's bookkeeping without the GPU engine
    s.prb = prb
    s._collect_constraints()
    return s


def test_srbd_problem_declares_the_references_functions_in_order():
    pb = SRBDProblem(); prb = pb.createSRBDProblem(NS, 1.0)
    cost = prb.function_container.getCost()
    assert list(cost) == (["rz_tracking", "o_tracking_xyz", "o_tracking_w", "rdot_tracking", "w_tracking", "rel_pos_y_1_4",      # prb.py:184-199
                           "rel_pos_x_1_4", "rel_pos_y_3_6", "rel_pos_x_3_6", "min_qddot"]                                         # :200
                          + [n for i in range(4) for n in (f"min_f{i}", f"f{i}_active")])                                          # :201-204
    for name, fn in cost.items():
        stage = name == "min_qddot" or name.startswith("min_f") or name.endswith("_active")
        assert fn.getNodes() == (list(range(0, NS)) if stage else list(range(1, NS + 1))), name
    cn = prb.function_container.getCnstr()
    assert list(cn) == (["relative_vel_left_1", "relative_vel_right_3"]                                                            # prb.py:166-170
                        + [n for i in range(4) for n in (f"cz_tracking{i}", f"cdotxy_tracking{i}")])                               # :179-181
    assert all(c.getNodes() == list(range(NS + 1)) for c in cn.values())                                                          # default: every node
    assert cost["min_qddot"].getDim() == 18 and cn["cdotxy_tracking2"].getDim() == 2


def test_equality_split_follows_the_reference_rule():
    pb = SRBDProblem(); prb = pb.createSRBDProblem(NS, 1.0)
    s = _adapter(prb)
    assert len(s.equality_constraints) == 10 and s.inequality_constraints == []       # the friction cone is commented out upstream
    assert all(s.is_equality_constraint(c) for c in s.equality_constraints)
    # with this build's opt-in barrier the cone is declared, as the inequality it is, and lands on the other side of the split
    pb2 = SRBDProblem(); prb2 = pb2.createSRBDProblem(NS, 1.0, params=dict(friction_barrier_weight=2.0))
    s2 = _adapter(prb2)
    assert [c.getName() for c in s2.inequality_constraints] == [f"f{i}_friction_cone" for i in range(4)]
    assert len(s2.equality_constraints) == 10
    assert s2.inequality_constraints[0].getNodes() == list(range(0, NS)) and np.all(np.isneginf(s2.inequality_constraints[0].getLowerBounds()))
    # ||ub - lb|| <= 1e-6 is the rule (ddp.py:108-111)
    c = prb.createConstraint("almost", Term("relative_vel", dim=2), bounds=dict(lb=[0.0, 0.0], ub=[5e-7, 5e-7]))
    assert s.is_equality_constraint(c)
    c = prb.createConstraint("loose", Term("relative_vel", dim=2), bounds=dict(lb=[0.0, 0.0], ub=[1e-6, 1e-6]))
    assert not s.is_equality_constraint(c)


@pytest.mark.parametrize("maker,model", [(lambda: SRBDProblem().createSRBDProblem(NS, 1.0), "srbd37"),
                                         (lambda: SRBD13Problem().createSRBD13Problem(NS, 1.0), "srbd13"),
                                         (lambda: LIPProblem().createLIPProblem(NS, 1.0), "lip30")])
def test_declared_functions_reproduce_the_default_model_constants(maker, model):
    prb = maker()
    s = _adapter(prb)
    consts = s._model_consts_from_functions()
    for k, v in prb.model_consts.items():
        np.testing.assert_array_equal(np.asarray(consts[k]), np.asarray(v), err_msg=k)
    assert set(prb.function_container.getCost()) == set(MODEL_TERMS[model]["cost"])
    assert sorted(c.getName() for c in s.equality_constraints) == sorted(MODEL_TERMS[model]["eq"])


def test_changed_costs_change_the_model_constants_or_raise():
    pb = SRBD13Problem(); prb = pb.createSRBD13Problem(NS, 1.0)
    prb.removeCostFunction("w_tracking")                                                   # a term switched off
    prb.createResidual("rz_tracking", Term("rz_tracking", "r_tracking_gain", 5e3), nodes=range(1, NS + 1))    # a changed weight
    consts = _adapter(prb)._model_consts_from_functions()
    assert consts["w_tracking_gain"] == 0.0 and consts["r_tracking_gain"] == 5e3 and consts["rdot_tracking_gain"] == 1e4
    # what the analytic model cannot express is refused, never dropped silently
    prb.createResidual("my_new_cost", Term("something_else"))
    with pytest.raises(NotImplementedError):
        _adapter(prb)._model_consts_from_functions()
    prb.removeCostFunction("my_new_cost")
    prb.createResidual("rdot_tracking", Term("rdot_tracking", "rdot_tracking_gain", 1e4, 3), nodes=range(0, NS))   # other nodes
    with pytest.raises(NotImplementedError):
        _adapter(prb)._model_consts_from_functions()
    pb = SRBDProblem(); prb = pb.createSRBDProblem(NS, 1.0)
    prb.createResidual("min_f2", Term("min_f", "min_f_gain", 0.5, 3), nodes=range(0, NS))   # one gain for the four forces
    with pytest.raises(NotImplementedError):
        _adapter(prb)._model_consts_from_functions()
    pb = SRBDProblem(); prb = pb.createSRBDProblem(NS, 1.0)
    prb.removeConstraint("cz_tracking1")                                                    # penalties are hard-wired
    with pytest.raises(NotImplementedError):
        _adapter(prb)._model_consts_from_functions()
    prb.createConstraint("cz_tracking1", Term("cz_tracking"))
    prb.createConstraint("box", Term("state_box", dim=3), bounds=dict(lb=-1.0, ub=1.0))      # an inequality without a barrier
    with pytest.raises(NotImplementedError):
        _adapter(prb)._model_consts_from_functions()


def test_pyddp_shaped_module_checks_its_arguments():
    from srbd_horizon_amd import pyddp_hip as pyddp
    o = pyddp.DdpSolverOptions()
    assert (o.max_iters, o.alpha_0, o.alpha_converge_threshold, o.line_search_decrease_factor, o.beta) == (100, 1.0, 1e-1, 0.5, 1e-4)
    f_list, L_list, L_term = pyddp.model_functions("srbd13", 30)
    assert len(f_list) == len(L_list) == 30 and f_list[3].name() == "f3" and L_term.name() == "L30"
    with pytest.raises(ValueError):
        pyddp.DdpSolver(37, 24, f_list, L_list, L_term, o)          # nx / nu of another model
    with pytest.raises(TypeError):
        pyddp.DdpSolver(13, 6, [lambda x, u: x] * 30, L_list, L_term, o)
    with pytest.raises(ValueError):
        pyddp.DdpSolver(13, 6, f_list[:-1], L_list, L_term, o)


def test_four_point_feet_is_the_srbd37_layout_without_relative_velocity_constraints():
    """number_of_legs = 4 x contact_model = 1 runs in the reference (prb.py:39-41: nc = 4) and declares NO relative_vel_*
    constraint (prb.py:166: `if contact_model > 1`).  Here: the srbd37 model with its runtime switch off -- through the builder
    and, equally, by removing those constraints from a contact_model = 2 problem; a partial set raises."""
    pb = SRBDProblem(); prb = pb.createSRBDProblem(NS, 1.0, params=dict(number_of_legs=4, contact_model=1))
    assert prb.model == "srbd37" and pb.nc == 4 and pb.contact_model == 1
    names = list(prb.function_container.getCnstr())
    assert not any(n.startswith("relative_vel_") for n in names) and len(names) == 8       # cz_tracking_i, cdotxy_tracking_i
    assert prb.model_consts["relative_velocity_constraints"] == 0
    c = _adapter(prb)._model_consts_from_functions()
    assert c["relative_velocity_constraints"] == 0
    assert pb.getInitialState().shape == (37,) and pb.getStaticInput().shape == (24,)      # prb.py:224-246 literals
    # the launch file's configuration keeps them
    pb2 = SRBDProblem(); prb2 = pb2.createSRBDProblem(NS, 1.0)
    assert _adapter(prb2)._model_consts_from_functions()["relative_velocity_constraints"] == 1
    # ... and dropping them from its container by hand is the same switch
    for n in ("relative_vel_left_1", "relative_vel_right_3"):
        prb2.removeConstraint(n)
    assert _adapter(prb2)._model_consts_from_functions()["relative_velocity_constraints"] == 0
    pb3 = SRBDProblem(); prb3 = pb3.createSRBDProblem(NS, 1.0)
    prb3.removeConstraint("relative_vel_left_1")                                           # one of two: no such model
    with pytest.raises(NotImplementedError):
        _adapter(prb3)._model_consts_from_functions()
    for bad in (dict(number_of_legs=4, contact_model=2), dict(number_of_legs=1, contact_model=4), dict(number_of_legs=2, contact_model=3)):
        with pytest.raises(ValueError):
            SRBDProblem().createSRBDProblem(NS, 1.0, params=bad)


def test_linear_terms_become_extra_rows_of_the_model():
    """A user adds a tracking term to prb.py's problem (ddp.py:183-196 would just sum it): problem.LinearTerm -> the extra rows
    of the model's "_x" build -- coefficient vector over z = [x u] in creation order, kind from the node range, the reference
    parameter mapped to the widened parameter vector.  What the analytic models cannot express still raises."""
    from srbd_horizon_amd.problem import LinearTerm
    pb = SRBDProblem(); prb = pb.createSRBDProblem(NS, 1.0)
    c0_ref = prb.createParameter("c0_xy_ref", 2)
    c0_ref.assign(np.array([0.1, 0.2]))
    c0_ref.assign(np.array([0.3, 0.4]), nodes=[NS])
    prb.createResidual("c0_xy_tracking", LinearTerm({pb.c[0]: [[1, 0, 0], [0, 1, 0]]}, gain=250.0, ref=c0_ref), nodes=range(1, NS + 1))
    prb.createResidual("f_balance", LinearTerm({pb.f[0]: [[0, 0, 1]], pb.f[2]: [[0, 0, -1]]}, gain=4.0, const=0.5), nodes=range(0, NS))
    a = _adapter(prb)
    a.state_var = prb.getState().getVars(); a.input_var = prb.getInput().getVars()
    a.state_size, a.input_size = 37, 24
    c = a._model_consts_from_functions()
    rows = c["extra_rows"]
    assert [r["kind"] for r in rows] == ["state", "state", "stage"] and [r["w"] for r in rows] == [250.0, 250.0, 4.0]
    assert rows[0]["a"][7] == 1.0 and rows[1]["a"][8] == 1.0 and np.count_nonzero(rows[0]["a"]) == 1          # c0 = x[7:10]
    assert rows[2]["a"][37 + 5] == 1.0 and rows[2]["a"][37 + 17] == -1.0 and rows[2]["const"] == 0.5          # f_i = u[6 i + 3 : 6 i + 6]
    a._np_model = 19
    P = a._parameter_matrix()
    assert P.shape == (NS + 1, 27)
    np.testing.assert_array_equal(P[:, :19], prb.parameter_matrix()[:, :19])
    np.testing.assert_array_equal(P[0, 19:22], [0.1, 0.2, 0.0]); np.testing.assert_array_equal(P[NS, 19:21], [0.3, 0.4])
    # limits: node ranges, inputs in a state term, more than 8 rows, non-linear anything
    prb.createResidual("bad_nodes", LinearTerm({pb.c[0]: [[1, 0, 0]]}, gain=1.0), nodes=range(2, NS))
    with pytest.raises(NotImplementedError):
        a._model_consts_from_functions()
    prb.removeCostFunction("bad_nodes")
    prb.createResidual("bad_inputs", LinearTerm({pb.f[0]: [[1, 0, 0]]}, gain=1.0), nodes=range(1, NS + 1))
    with pytest.raises(NotImplementedError):
        a._model_consts_from_functions()
    prb.removeCostFunction("bad_inputs")
    prb.createResidual("too_many", LinearTerm({pb.c[1]: np.eye(3), pb.c[2]: np.eye(3)}, gain=1.0), nodes=range(1, NS + 1))
    prb.createResidual("too_many2", LinearTerm({pb.cdot[1]: np.eye(3)}, gain=1.0), nodes=range(1, NS + 1))
    with pytest.raises(NotImplementedError):
        a._model_consts_from_functions()
    with pytest.raises(ValueError):
        LinearTerm({pb.c[0]: [[1, 0]]}, gain=1.0)
