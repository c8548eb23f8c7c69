// covid19.hpp -- host mirror of the reference's SEIR epidemic model class (covid19.hpp:17-88);
// dynamics on the device (SOCP_MODEL_COVID19).
#ifndef SOCP_AMD_COVID19_HPP_
#define SOCP_AMD_COVID19_HPP_

#include <iostream>   // user programs written for the reference rely on these transitive includes

#include "../../socp/map.hpp"
#include "../../socp/model.hpp"

class covid19 : public model
{
    struct data_struct;
    data_struct *data;

    // the plugin virtuals: evaluated by the device twin
    mstate Model(real const &t, mstate const &X, int isJac = 0) const override;
    mcontrol Control(real const &t, mstate const &X) const override;
    mstate Hamiltonian(real const &t, mstate const &X, int isJac = 0) const override;
    mstate ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac = 0) override;

public:
    // same members, order and types as the reference's structure (users assign them through GetParameterData())
    struct parameters_struct {
        real R0, Tinf, Tinc;   // secondary infections per case; infectious period; incubation period
        real N;                // population size (1 when normalised)
        real Imax, muI;        // tolerated infectious fraction and the penalty weight beyond it
        real umin, umax;       // control bounds
    };

    covid19(std::string the_fileTrace = std::string(""));
    ~covid19() override;
    parameters_struct &GetParameterData();

    // device hook
    int DeviceModelId() const override;
    int DeviceParams(double *out, int cap) const override;
    int DeviceStepNumber() const override;
};

#endif
