// goddard.hpp -- host mirror of the reference's 3-D Goddard rocket model class (goddard.hpp:19-112).
// Same constructor, parameter names and public methods; the dynamics, control law and Hamiltonian
// are evaluated by their gfx950 twins (socp_amd/csrc/models_exact.hpp, SOCP_MODEL_GODDARD).
#ifndef SOCP_AMD_GODDARD_HPP_
#define SOCP_AMD_GODDARD_HPP_

#include <iostream>   // user programs written for the reference rely on these transitive includes

#include "../../socp/map.hpp"
#include "../../socp/model.hpp"

class goddard : public model
{
    struct data_struct;
    data_struct *data;

    mstate Model(real const &t, mstate const &X, int isJac) const override;
    mcontrol Control(real const &t, mstate const &X) const override;
    mstate Hamiltonian(real const &t, mstate const &X, int isJac) const override;
    mstate ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac) override;
    void Trace(real const &t, mstate const &X, std::stringstream &file) const override;
    real GetSingularControl(real t, mstate const &X) const;

public:
    // default parameter values (goddard.hpp:28-37); the live values are the entries of model::parameters
    struct parameters_struct {
        real C = 3.5, b = 7.0;         // thrust and mass-flow coefficients
        real KD = 310.0, kr = 500.0;   // drag coefficient, air-density scale
        real u_max = 1.0;              // control bound
        real mu1 = 1.0, mu2 = 0.0;     // weights of |u| and |u|^2 in the cost (mu2 > 0: smooth control law)
        real singularControl = -1;     // < 0: closed-form singular arc, else constant value
    };

    goddard(std::string the_fileTrace = std::string(""), int stepNbr = 10);
    ~goddard() override;

    mstate SwitchingTimesFunction(real const &t, mstate const &X, mstate const &Xp, int isJac) const override;
    void SwitchingTimesUpdate(std::vector<real> const &switchingTimes) override;
    void SetParameterDataName(std::string name, real value);
    real &GetParameterDataName(std::string name);

    // device hook
    int DeviceModelId() const override;
    int DeviceParams(double *out, int cap) const override;
    std::vector<real> DeviceSwitchingTimes() const override;
};

#endif
