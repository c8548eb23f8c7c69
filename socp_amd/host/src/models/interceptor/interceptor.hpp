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
public:
    // interceptor.hpp:28-46
    struct parameters_struct {
        real c0;                 // max curvature at ground level (1/m)
        real hr;                 // reference altitude (m)
        real d0;                 // drag at ground level (1/m)
        real eta;                // coefficient of efficiency
        real propellant_mass;    // (kg)
        real empty_mass;         // (kg)
        real q;                  // mass flow rate (kg/s)
        real ve;                 // gas speed (m/s)
        real alpha_max;          // max angle of attack (rad)
        real u_max;              // bound of the normalised control
        real a_max;              // max acceleration (unused by the dynamics)
        real r_2p;               // declared by the reference, never read
        real t_2p;               // declared by the reference, never read
        real mu_gft;             // 0..1 homotopy on gravity and thrust
        real muT;                // weight of time in the cost
        real muV;                // weight of final velocity in the cost
        real muC;                // weight of the quadratic control cost
    };

    interceptor(std::string the_fileTrace = std::string(""));
    virtual ~interceptor();

    parameters_struct &GetParameterData();

    // two stages (powered until propellant_mass/q, then coasting), chart re-chosen before every step, state
    // returned in chart 1 (interceptor.cpp:162-218)
    virtual mstate ComputeTraj(real const &t0, mstate const &X0, real const &tf, int isTrace, int isJac);

    virtual void FinalFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const;
    virtual void FinalHFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const;

    // closed-form costate guess (IFAC WC 2017 paper cited by the reference); fills Xi[6..12)
    void InitAnalytical(real const &ti, mstate &Xi, real const &tf, mstate &Xf) const;

    // device hook
    virtual int DeviceModelId() const;
    virtual int DeviceParams(double *out, int cap) const;
    virtual int DeviceStepNumber() const;
    virtual std::vector<real> DeviceSwitchingTimes() const;     // (stageMode, currentChart): the device twin's two flags

private:
    struct data_struct;
    data_struct *data;

    virtual mstate Model(real const &t, mstate const &X, int isJac) const;
    virtual mcontrol Control(real const &t, mstate const &X) const;
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac) const;
    virtual int GetMode(real const &t, mstate const &X) const;
    real ComputeMass(real const &t, mstate const &X) const;
};

#endif
