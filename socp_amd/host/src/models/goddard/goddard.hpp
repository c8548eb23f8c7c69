// goddard.hpp -- host mirror of the reference's 3-D Goddard rocket model class (goddard.hpp:19-112).
// Same constructor, parameter names and public methods; the dynamics, control law and Hamiltonian
// are evaluated by their gfx950 twins (socp_amd/csrc/models_exact.hpp, SOCP_MODEL_GODDARD).
#ifndef SOCP_AMD_GODDARD_HPP_
#define SOCP_AMD_GODDARD_HPP_

#include "../../socp/model.hpp"
#include "../../socp/map.hpp"

#include <iostream>

class goddard : public model
{
public:
    // default parameter values (goddard.hpp:28-37)
    struct parameters_struct {
        real C = 3.5;                  // thrust coefficient
        real b = 7.0;                  // mass flow coefficient
        real KD = 310.0;               // drag coefficient
        real kr = 500.0;               // air density scale
        real u_max = 1.0;              // control bound
        real mu1 = 1.0;                // weight of |u| in the cost
        real mu2 = 0.0;                // weight of |u|^2 in the cost (> 0: smooth control law)
        real singularControl = -1;     // < 0: closed-form singular arc, else constant value
    };

    goddard(std::string the_fileTrace = std::string(""), int stepNbr = 10);
    virtual ~goddard();

    virtual mstate SwitchingTimesFunction(real const &t, mstate const &X, mstate const &Xp, int isJac) const;
    virtual void SwitchingTimesUpdate(std::vector<real> const &switchingTimes);
    void SetParameterDataName(std::string name, real value);
    real &GetParameterDataName(std::string name);

    // device hook
    virtual int DeviceModelId() const;
    virtual int DeviceParams(double *out, int cap) const;
    virtual std::vector<real> DeviceSwitchingTimes() const;

private:
    struct data_struct;
    data_struct *data;

    virtual mstate Model(real const &t, mstate const &X, int isJac) const;
    virtual mcontrol Control(real const &t, mstate const &X) const;
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac) const;
    virtual mstate ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac);
    virtual void Trace(real const &t, mstate const &X, std::stringstream &file) const;
    real GetSingularControl(real t, mstate const &X) const;
};

#endif
