// interceptor.hpp -- host mirror of the reference's endo-atmospheric interceptor model class
// (interceptor.hpp:21-143).  Same constructor, parameter structure and public methods; the two-chart
// dynamics, the control law, the Hamiltonian, the two-stage trajectory with chart switching and the
// final-boundary rows are evaluated by the gfx950 twin (socp_amd/csrc/models_interceptor.hpp,
// SOCP_MODEL_INTERCEPTOR).  InitAnalytical -- a closed-form guess computed once per problem -- is host
// arithmetic.
#ifndef SOCP_AMD_INTERCEPTOR_HPP_
#define SOCP_AMD_INTERCEPTOR_HPP_

#include "../../socp/model.hpp"
#include "../../socp/map.hpp"

#include <iostream>

class interceptor : public model
{
    struct data_struct;
    data_struct *data;

    // evaluated by the device twin in the chart / stage the object is currently in
    mstate Model(real const &t, mstate const &X, int isJac) const override;
    mcontrol Control(real const &t, mstate const &X) const override;
    mstate Hamiltonian(real const &t, mstate const &X, int isJac) const override;
    int GetMode(real const &t, mstate const &X) const override;
    real ComputeMass(real const &t, mstate const &X) const;

public:
    // same members, order and types as interceptor.hpp:28-46 (assigned through GetParameterData())
    struct parameters_struct {
        real c0, hr, d0;                      // max curvature and drag at ground level (1/m), reference altitude (m)
        real eta;                             // coefficient of efficiency
        real propellant_mass, empty_mass;     // (kg)
        real q, ve;                           // mass flow rate (kg/s), gas speed (m/s)
        real alpha_max, u_max;                // max angle of attack (rad), bound of the normalised control
        real a_max;                           // max acceleration (not used by the dynamics)
        real r_2p, t_2p;                      // declared by the reference, never read
        real mu_gft;                          // 0..1 homotopy on gravity and thrust
        real muT, muV, muC;                   // cost weights: time, final velocity, quadratic control
    };

    interceptor(std::string the_fileTrace = std::string(""));
    ~interceptor() override;

    parameters_struct &GetParameterData();

    // two stages (powered until propellant_mass/q, then coasting), chart re-chosen before every step, state
    // returned in chart 1 (interceptor.cpp:162-218)
    mstate ComputeTraj(real const &t0, mstate const &X0, real const &tf, int isTrace, int isJac) override;

    void FinalFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const override;
    void FinalHFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const override;

    // closed-form costate guess (IFAC WC 2017 paper cited by the reference); fills Xi[6..12)
    void InitAnalytical(real const &ti, mstate &Xi, real const &tf, mstate &Xf) const;

    // device hook
    int DeviceModelId() const override;
    int DeviceParams(double *out, int cap) const override;
    int DeviceStepNumber() const override;
    std::vector<real> DeviceSwitchingTimes() const override;     // (stageMode, currentChart): the device twin's two flags
};

#endif
